#!/bin/bash
# Experiment builds: tools/build_variant.sh <tag> "<extra hipcc flags>" [kinds...]
# recompiles only the f64 / KIND 0 translation unit of the step kernels (the one every benchmark config runs) with the extra
# flags (on top of the Makefile's KFLAGS; TDS_KFLAGS overrides those) into build/obj_x<tag>/ and links it with the standard
# objects of build/obj/ into libtds_hip_x<tag>.so
# (loaded with TDS_HIP_LIB=... by the python binding; never shipped as the default library).
set -e
TAG=$1; EXTRA=$2
cd "$(dirname "$0")/../tiny-differentiable-simulator_amd/csrc"
ROOT=$(cd ../.. && pwd)
G=$ROOT/build/obj_x$TAG
mkdir -p $G
KF=${TDS_KFLAGS--mllvm -disable-machine-licm}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I. -Wall -Wno-unused-function $KF $EXTRA \
  -DTDS_ONLY_F64 -DTDS_ONLY_KIND=0 -c -o $G/tds_kernels_f64_k0.o tds_kernels.hip
OBJS=$(ls $ROOT/build/obj/*.o | grep -v tds_kernels_f64_k0.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../libtds_hip_x$TAG.so $G/tds_kernels_f64_k0.o $OBJS -ldl
ls -la ../libtds_hip_x$TAG.so
