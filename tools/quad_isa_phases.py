#!/usr/bin/env python3
"""Static instruction counts of the 16-lane kernel between its phase stamps (tds_quad.hip built with -DTDS_QUAD_PROF: every
QUAD_STAMP is one s_memtime in the assembly), per kernel instantiation: what each phase of tools/quad_profile.py's cycle table
issues (loop bodies — the rows of a contact window — counted once).
usage: tools/quad_isa_phases.py [kernel substring, default IddLb1ELi8 = f64 step-loop form in wide workgroups]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "tiny-differentiable-simulator_amd", "csrc")
PH = ["A prologue / top of the step", "PD", "B jcalc", "C root chain", "leg scans", "D rigid inertia", "I narrowphase + count", "M1 visual poses + root inertia",
      "E composites", "G rows of M", "H LDL^T", "F forward dynamics", "J K L rows + sweep + impulse", "M integrate", "y record",
      "N reward / done", "reset pool", "obs record", "(behind the last stamp)"]


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "IddLb1ELi8"
    out = "/tmp/quad_prof.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CS,
                           "-Wno-unused-function", "-mllvm", "-disable-machine-licm", "-DTDS_QUAD_PROF", "--cuda-device-only", "-S", "-o", out,
                           os.path.join(CS, "tds_quad.hip")], stderr=subprocess.DEVNULL)
    txt = open(out).read()
    m = re.search(r"^(_Z\w*tds_quad_kernel" + want + r"\w*):.*?\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M)
    body = m.group(2).split("\n")
    k, counts = 0, collections.OrderedDict()
    for l in body:
        t = l.strip()
        if not l.startswith("\t") or not t or t.startswith((".", ";")):
            continue
        op = t.split()[0]
        if op == "s_memtime":
            k += 1
            continue
        c = counts.setdefault(k, collections.Counter())
        c["total"] += 1
        c["valu"] += op.startswith("v_")
        c["f64"] += op.startswith("v_") and "f64" in op
        c["ds"] += op.startswith("ds_")
        c["salu"] += op.startswith("s_")
        c["mem"] += op.startswith(("global_", "flat_", "scratch_"))
        c["dpp"] += ("dpp" in t or "quad_perm" in t or "row_" in t)
        c["wait"] += op == "s_waitcnt"
    print(m.group(1))
    print(f"{'phase':34s} {'total':>6s} {'valu':>6s} {'f64':>5s} {'dpp':>5s} {'ds':>5s} {'salu':>5s} {'mem':>4s} {'waits':>5s}")
    tot = collections.Counter()
    for k, c in counts.items():
        tot.update(c)
        print(f"{PH[k] if k < len(PH) else k:34s} {c['total']:6d} {c['valu']:6d} {c['f64']:5d} {c['dpp']:5d} {c['ds']:5d} {c['salu']:5d} {c['mem']:4d} {c['wait']:5d}")
    print(f"{'sum':34s} {tot['total']:6d} {tot['valu']:6d} {tot['f64']:5d} {tot['dpp']:5d} {tot['ds']:5d} {tot['salu']:5d} {tot['mem']:4d} {tot['wait']:5d}")


if __name__ == "__main__":
    main()
