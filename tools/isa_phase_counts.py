#!/usr/bin/env python3
"""Static instruction counts of ONE step-kernel instantiation per phase, from the device assembly (no GPU needed).

The phase-stamp builds (PROF = true) read the shader clock at every phase boundary (TDS_STAMP: s_memtime), so the
instructions between two consecutive s_memtime of such a build are the instructions of one phase.  Loops (tree levels,
contact sweeps) are counted once — this is what the wavefront FETCHES, not what it executes — but in an issue-bound kernel
(config 4: two wavefronts per SIMD at 60 % of the VALU's issue slots) it shows where instructions can be taken out, which
the executed counts of the SQ counters (profiles/*_sq_counters.txt) cannot attribute to a phase.

usage: python tools/isa_phase_counts.py <lanes*100+ndp> [extra hipcc flags ...]
   e.g. python tools/isa_phase_counts.py 3218            (Laikago: 32 lanes, 18 dof)
        python tools/isa_phase_counts.py 1614 -DTDS_X    (the Ant with an experiment macro)
Prints, for the one-wavefront phase-stamp build <double, double, G, NDP, true, 0, 0, false>, VALU / SALU / LDS / VMEM
counts per phase with the share of f64 arithmetic, DPP moves, selects, register copies and s_nop in the VALU count, and
the op mix of the straight-line build <..., false, 0, 0, false> the graphs launch."""
import collections
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "tiny-differentiable-simulator_amd", "csrc")
key = int(sys.argv[1]) if len(sys.argv) > 1 else 3218
extra = sys.argv[2:]
G, NDP = key // 100, key % 100
asm = os.path.join(tempfile.gettempdir(), f"tds_k{key}.s")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{CS}",
       "-mllvm", "-disable-machine-licm", "-DTDS_ONLY_F64", "-DTDS_ONLY_KIND=0", f"-DTDS_DEBUG_ONLY={key}",
       "--cuda-device-only", "-S", "-o", asm, os.path.join(CS, "tds_kernels.hip")] + extra
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")


def body(prof, loop, w2):
    tag = f"tds_step_kernelIddLi{G}ELi{NDP}ELb{int(prof)}ELi{loop}ELi0ELb{int(w2)}EEE"
    a = next(i for i, l in enumerate(lines) if l.startswith("_Z") and tag in l.split(":")[0])
    b = next(i for i in range(a + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[a + 1:b]


def classify(op, text, c):
    if op.startswith("v_"):
        c["valu"] += 1
        if "_f64" in op and not op.startswith(("v_cmp", "v_cvt")):
            c["f64"] += 1
        if "dpp" in text or "permlane" in op:
            c["dpp"] += 1
        if op.startswith("v_cndmask"):
            c["select"] += 1
        if op in ("v_mov_b32_e32", "v_mov_b64_e32"):
            c["mov"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        c["vmem"] += 1
    elif op == "s_nop":
        c["nop"] += 1
    elif op == "s_waitcnt":
        c["wait"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1


names = ["(prologue)", "A load + PD", "B jcalc", "C kinematics", "I narrowphase + M1 + D inertias", "E composite sweep", "G mass matrix",
         "H LDLt", "F forward-dynamics solve", "(barrier)", "J jacobian rows", "K row solves", "L contact solve", "M/N integrate + pack",
         "(epilogue)"]
segs = [collections.Counter()]
for l in body(True, 0, False):
    t = l.strip()
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_memtime":
        segs.append(collections.Counter())
        continue
    classify(op, t, segs[-1])
print(f"<double, double, {G}, {NDP}, PROF, 0, 0, one wavefront>: static instructions per phase" + (f"  [{' '.join(extra)}]" if extra else ""))
print(f"{'phase':34s} {'VALU':>5s} {'f64':>5s} {'dpp':>5s} {'sel':>5s} {'mov':>5s} {'SALU':>5s} {'LDS':>5s} {'VMEM':>5s} {'nop':>4s} {'wait':>5s}")
tot = collections.Counter()
for i, c in enumerate(segs):
    tot.update(c)
    print(f"{names[i] if i < len(names) else '?':34s} {c['valu']:5d} {c['f64']:5d} {c['dpp']:5d} {c['select']:5d} {c['mov']:5d} {c['salu']:5d} {c['lds']:5d} {c['vmem']:5d} {c['nop']:4d} {c['wait']:5d}")
print(f"{'total':34s} {tot['valu']:5d} {tot['f64']:5d} {tot['dpp']:5d} {tot['select']:5d} {tot['mov']:5d} {tot['salu']:5d} {tot['lds']:5d} {tot['vmem']:5d} {tot['nop']:4d} {tot['wait']:5d}")
ops = collections.Counter()
for l in body(False, 0, False):
    t = l.strip()
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    ops[t.split()[0]] += 1
print(f"\nstraight-line build <double, double, {G}, {NDP}, false, 0, 0, false>: {sum(v for k, v in ops.items() if k.startswith('v_'))} VALU of {sum(ops.values())} instructions; most frequent:")
print("  " + ", ".join(f"{k} {v}" for k, v in ops.most_common(24)))
