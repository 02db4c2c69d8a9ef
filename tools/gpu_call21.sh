#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/c21
mkdir -p $O
for at in 1 2 3; do
TDS_GRAM_STAMP_AT=$at timeout 200 python tools/profile_phases.py ant 4096 0 100 > $O/phases_at$at.txt 2>&1
echo "at=$at"; grep -A16 "  main wavefront" $O/phases_at$at.txt | sed -n 10,16p
done
