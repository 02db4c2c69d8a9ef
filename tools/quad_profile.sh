#!/bin/bash
# Phase stamps of the 16-lane kernel (csrc/tds_quad.hip, -DTDS_QUAD_PROF): builds tds_quad.hip once more with the stamps and links
# it with the library's other objects into libtds_hip_quadprof.so (no GPU needed; run here), which tools/quad_profile.py loads
# on the GPU box:   tools/quad_profile.sh && gpurun -- 'python tools/quad_profile.py > gpurun_out/quad_phases.txt'
set -e
cd "$(dirname "$0")/../tiny-differentiable-simulator_amd/csrc"
OBJ=../../build/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -mllvm -disable-machine-licm \
  --offload-compress -DTDS_QUAD_PROF ${QUAD_PROF_EXTRA} -c -o $OBJ/tds_quad_prof.o tds_quad.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../libtds_hip_quadprof.so \
  $(ls $OBJ/tds_kernels_*.o) $OBJ/tds_api.o $OBJ/tds_shard.o $OBJ/tds_rb.o $OBJ/tds_quad_prof.o $OBJ/tds_chain.o $OBJ/tds_oct.o -ldl
ls -la ../libtds_hip_quadprof.so
