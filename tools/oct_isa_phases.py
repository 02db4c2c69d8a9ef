#!/usr/bin/env python3
"""Instructions between the OCTMARK comments of the 8-lane kernel's assembly (built with -DTDS_OCT_MARKS), per kernel: on a lone
wavefront an instruction costs ~5 cycles whatever it depends on (tools/ubench/lone_wave_latency.hip), so these counts are the
phase times.  usage: tools/oct_isa_phases.py [kernel substring, default IddLb1ELi3 = f64 step-loop two-wavefront build]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "tiny-differentiable-simulator_amd", "csrc")


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "IddLb1ELi3"
    out = "/tmp/oct_marks.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CS,
                           "-Wno-unused-function", "-mllvm", "-disable-machine-licm", "-ffp-contract=on", "-DTDS_OCT_MARKS", "--cuda-device-only", "-S", "-o", out,
                           os.path.join(CS, "tds_oct.hip")], stderr=subprocess.DEVNULL)
    txt = open(out).read()
    m = re.search(r"^(_Z\w*tds_oct_kernel" + want + r"\w*):.*?\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M)
    body = m.group(2).split("\n")
    cur, counts, order = "(prologue)", collections.OrderedDict(), []
    for l in body:
        t = l.strip()
        mm = re.match(r";\s*OCTMARK\s+(\S+)", t)
        if mm:
            cur = mm.group(1)
            continue
        if not l.startswith("\t") or not t or t.startswith((".", ";")):
            continue
        op = t.split()[0]
        c = counts.setdefault(cur, collections.Counter())
        c["total"] += 1
        c["valu"] += op.startswith("v_")
        c["ds"] += op.startswith("ds_")
        c["salu"] += op.startswith("s_")
        c["mem"] += op.startswith(("global_", "flat_", "scratch_"))
        c["dpp"] += ("dpp" in t or "quad_perm" in t or "row_" in t)
        c["wait"] += op == "s_waitcnt"
    print(m.group(1))
    print(f"{'after mark':22s} {'total':>6s} {'valu':>6s} {'ds':>5s} {'salu':>5s} {'mem':>4s} {'dpp':>4s} {'waits':>5s}")
    for k, c in counts.items():
        print(f"{k:22s} {c['total']:6d} {c['valu']:6d} {c['ds']:5d} {c['salu']:5d} {c['mem']:4d} {c['dpp']:4d} {c['wait']:5d}")


if __name__ == "__main__":
    main()
