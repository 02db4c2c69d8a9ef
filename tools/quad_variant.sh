#!/bin/bash
# Experiment builds of the 16-lane kernel: tools/quad_variant.sh <tag> "<extra hipcc flags>" -> libtds_hip_q<tag>.so (tds_quad.hip
# recompiled with the flags, linked with the library's other objects; loaded with TDS_HIP_LIB=... or tools/ab_libs.sh)
set -e
TAG=$1; EXTRA=$2
cd "$(dirname "$0")/../tiny-differentiable-simulator_amd/csrc"
OBJ=../../build/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -mllvm -disable-machine-licm \
  --offload-compress $EXTRA -c -o $OBJ/tds_quad_q$TAG.o tds_quad.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../libtds_hip_q$TAG.so \
  $(ls $OBJ/tds_kernels_*.o) $OBJ/tds_api.o $OBJ/tds_shard.o $OBJ/tds_rb.o $OBJ/tds_quad_q$TAG.o $OBJ/tds_chain.o $OBJ/tds_oct.o -ldl
rm -f $OBJ/tds_quad_q$TAG.o
ls -la ../libtds_hip_q$TAG.so
