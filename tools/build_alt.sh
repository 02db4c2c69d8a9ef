#!/bin/bash
# Experiment slots (csrc/tds_kernels.h): tools/build_alt.sh <lanes*100+ndp> "<flags of slot 1>" ["<flags of slot 2>" ...]
# compiles the f64 / KIND 0 step kernels of ONE instantiation (1614: the Ant, 3218: Laikago) once per slot with the slot's
# extra flags — on top of the Makefile's KFLAGS unless the slot's flags start with "NOKFLAGS" — into build/obj/tds_alt<k>.o
# and relinks libtds_hip.so WITH the slots (bench.py --option alt_build=k / HipSim(options={"alt_build": k}) selects one).
# `make -C tiny-differentiable-simulator_amd/csrc lib` afterwards (or any change of the standard objects) links the
# library without them again.
set -e
KEY=$1; shift
cd "$(dirname "$0")/../tiny-differentiable-simulator_amd/csrc"
ROOT=$(cd ../.. && pwd)
G=$ROOT/build/obj
mkdir -p $G
rm -f $G/tds_alt*.o
k=0
pids=""
for FL in "$@"; do
  k=$((k+1))
  KF="-mllvm -disable-machine-licm"
  case "$FL" in NOKFLAGS*) KF=""; FL="${FL#NOKFLAGS}";; esac
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I. -Wall -Wno-unused-function $KF $FL \
      -DTDS_ONLY_F64 -DTDS_ONLY_KIND=0 -DTDS_DEBUG_ONLY=$KEY -DTDS_ALT=$k -Rpass-analysis=kernel-resource-usage \
      -c -o $G/tds_alt$k.o tds_kernels.hip > $G/tds_alt$k.log 2>&1 || { echo "slot $k FAILED"; tail -5 $G/tds_alt$k.log; } ) &
  pids="$pids $!"
done
wait $pids
for i in $(seq 1 $k); do
  echo "slot $i: $(eval echo \${$i})"
  python3 - $G/tds_alt$i.log <<'P'
import re,sys,subprocess
cur=None; rows=[]
for line in open(sys.argv[1]):
    m=re.search(r"Function Name: (\S+)",line)
    if m:
        cur={"name":m.group(1)}; rows.append(cur); continue
    m=re.search(r"remark:\s+([\w \[\]/]+?): (\d+)",line)
    if m and cur is not None: cur[m.group(1).strip()]=int(m.group(2))
for r in rows:
    d=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    m=re.search(r"tds_step_kernel<(.*?)>\(",d)
    print("   %-62s VGPR %3d AGPR %3d scratch %4d occ %d"%((m.group(1) if m else d[:60]), r.get("VGPRs",-1), r.get("AGPRs",-1), r.get("ScratchSize [bytes/lane]",-1), r.get("Occupancy [waves/SIMD]",-1)))
P
done
OBJS=$(ls $G/*.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../libtds_hip.so $OBJS -ldl
ls -la ../libtds_hip.so
