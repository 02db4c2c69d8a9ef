#!/bin/bash
# Runs ON THE GPU BOX: 1000-step closed loops of both benchmark robots against the real reference at the round's final build.
export TMPDIR=/tmp
mkdir -p gpurun_out/profiles
{ echo "# tools/long_closed_loop_vs_reference.py 1000   (MI355X, round-4 final build; the REAL reference libtds_ref.so on the host threads,"; echo "# per-step resync: every step starts from the state the device held)"; timeout 500 python tools/long_closed_loop_vs_reference.py 1000 2>&1 | grep -v amdgpu.ids; } > gpurun_out/profiles/r04h_long_closed_loop_vs_reference.txt
cat gpurun_out/profiles/r04h_long_closed_loop_vs_reference.txt | cut -c1-250
