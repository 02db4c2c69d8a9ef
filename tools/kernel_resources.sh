#!/bin/bash
# Compiler-reported resources of every step-kernel instantiation (no GPU needed):
#   tools/kernel_resources.sh [f64|mix|f32] [kind] > profiles/rNN_kernel_resources_<build>_k<kind>.txt
# VGPR / AGPR / scratch / occupancy per instantiation from -Rpass-analysis=kernel-resource-usage, compiled with the
# Makefile's KFLAGS (TDS_KFLAGS= overrides them, TDS_EXTRA_FLAGS adds to them).
set -e
BUILD=${1:-f64}; KIND=${2:-0}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/tiny-differentiable-simulator_amd/csrc
DEF=$(echo $BUILD | tr a-z A-Z)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$CS -DTDS_ONLY_$DEF -DTDS_ONLY_KIND=$KIND \
  ${TDS_KFLAGS--mllvm -disable-machine-licm} $TDS_EXTRA_FLAGS -Rpass-analysis=kernel-resource-usage -c -o /dev/null $CS/tds_kernels.hip 2>&1 | python3 -c '
import re,sys
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m:
        cur={"name":m.group(1)}; rows.append(cur); continue
    m=re.search(r"remark:\s+([\w \[\]/]+?): (\d+)",line)
    if m and cur is not None: cur[m.group(1).strip()]=int(m.group(2))
import subprocess
print("%-74s %5s %5s %7s %4s %6s %6s %6s"%("kernel <T, TR, G, NDP, PROF, LOOP, KIND>","VGPR","AGPR","scratch","occ","sgprSp","vgprSp","SGPR"))
for r in rows:
    d=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    m=re.search(r"tds_step_kernel<(.*?)>\(",d)
    print("%-74s %5d %5d %7d %4d %6d %6d %6d"%(("tds_step_kernel<"+m.group(1)+">") if m else d[:74], r.get("VGPRs",-1), r.get("AGPRs",-1), r.get("ScratchSize [bytes/lane]",-1), r.get("Occupancy [waves/SIMD]",-1), r.get("SGPRs Spill",-1), r.get("VGPRs Spill",-1), r.get("TotalSGPRs",-1)))
'
