#!/bin/bash
# Phase stamps of the 8-lane kernel (csrc/tds_oct.hip, -DTDS_OCT_PROF): builds tds_oct.hip once more with the stamps and links
# it with the library's other objects into libtds_hip_octprof.so (no GPU needed; run here), which tools/oct_profile.py loads
# on the GPU box:   tools/oct_profile.sh && gpurun -- 'python tools/oct_profile.py > gpurun_out/oct_phases.txt'
set -e
cd "$(dirname "$0")/../tiny-differentiable-simulator_amd/csrc"
OBJ=../../build/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -mllvm -disable-machine-licm \
  -ffp-contract=on --offload-compress -DTDS_OCT_PROF ${OCT_PROF_EXTRA} -c -o $OBJ/tds_oct_prof.o tds_oct.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../libtds_hip_octprof.so \
  $(ls $OBJ/tds_kernels_*.o) $OBJ/tds_api.o $OBJ/tds_shard.o $OBJ/tds_rb.o $OBJ/tds_quad.o $OBJ/tds_chain.o $OBJ/tds_oct_prof.o -ldl
ls -la ../libtds_hip_octprof.so
