#!/bin/bash
# Compiler-reported resources of the kernels with a translation unit of their own (no GPU needed):
#   tools/special_kernel_resources.sh oct|quad|chain > profiles/rNN_<name>_kernel_resources.txt
# VGPR / AGPR / scratch / occupancy / spills per instantiation from -Rpass-analysis=kernel-resource-usage, compiled with the
# Makefile's flags for that unit.
set -e
WHICH=${1:?oct|quad|chain}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/tiny-differentiable-simulator_amd/csrc
FLAGS="-mllvm -disable-machine-licm"
[ "$WHICH" != quad ] && FLAGS="$FLAGS -ffp-contract=on"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$CS $FLAGS \
  -Rpass-analysis=kernel-resource-usage -c -o /dev/null $CS/tds_$WHICH.hip 2>&1 | python3 -c '
import re,sys,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m:
        cur={"name":m.group(1)}; rows.append(cur); continue
    m=re.search(r"remark:\s+([\w \[\]/]+?): (\d+)",line)
    if m and cur is not None: cur[m.group(1).strip()]=int(m.group(2))
print("%-64s %5s %5s %7s %4s %6s %6s %6s %7s"%("kernel","VGPR","AGPR","scratch","occ","sgprSp","vgprSp","SGPR","LDS"))
for r in rows:
    d=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    m=re.search(r"(tds_\w+_kernel\w*<.*?>)\(",d)
    print("%-64s %5d %5d %7d %4d %6d %6d %6d %7d"%(m.group(1) if m else d[:64], r.get("VGPRs",-1), r.get("AGPRs",-1), r.get("ScratchSize [bytes/lane]",-1), r.get("Occupancy [waves/SIMD]",-1), r.get("SGPRs Spill",-1), r.get("VGPRs Spill",-1), r.get("TotalSGPRs",-1), r.get("LDS Size [bytes/block]",-1)))
'
