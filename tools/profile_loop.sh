#!/bin/bash
# Phase stamps of a STEADY-STATE step of the two-wavefront step-loop kernel (Ant x 4096): iteration K / 2 of a K-step launch,
# from a profiling build of the f64 plain kernels (-DTDS_PROF_LOOP: csrc/tds_kernels.hip) linked into a library of its own —
# the shipped library has no such kernel.  Build HERE (no GPU needed), run on the GPU box:
#     tools/profile_loop.sh build         ->  ab_r05/libtds_hip_prof.so   (travels with the snapshot; remove it afterwards)
#     gpurun -- 'tools/profile_loop.sh run [K]'
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
case "${1:-run}" in
build)
  cd $ROOT/tiny-differentiable-simulator_amd/csrc
  G=$ROOT/build/prof; mkdir -p $G $ROOT/ab_r05
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I. -Wno-unused-function -mllvm -disable-machine-licm \
    -DTDS_ONLY_F64 -DTDS_ONLY_KIND=0 -DTDS_DEBUG_ONLY=1614 -DTDS_PROF_LOOP -c -o $G/tds_kernels_f64_k0.o tds_kernels.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I. -Wno-unused-function -c -o $G/tds_api.o tds_api.hip
  OBJS=$(ls $ROOT/build/obj/*.o | grep -v "tds_alt\|tds_kernels_f64_k0.o\|tds_api.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $ROOT/ab_r05/libtds_hip_prof.so $OBJS $G/tds_kernels_f64_k0.o $G/tds_api.o -ldl ;;
run)
  cd $ROOT
  K=${2:-200}
  echo "# steady-state step of the two-wavefront step-loop kernel: iteration $((K/2)) of a $K-step launch (no record rings: substeps)"
  TDS_HIP_LIB=$ROOT/ab_r05/libtds_hip_prof.so TDS_HIP_PROF_LOOP=$K python tools/profile_phases.py ant 4096 0 100 2>&1 | grep -v amdgpu.ids ;;  # (TDS_HIP_PROF_ITER=i: stamp iteration i instead of K / 2)
esac
