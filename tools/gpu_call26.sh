#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/c26
mkdir -p $O
for at in 11 12 13 14 15; do
TDS_GRAM_STAMP_AT=$at timeout 200 python tools/profile_phases.py ant 4096 0 100 > $O/phases_at$at.txt 2>&1
echo "at=$at: $(grep -A14 '  main wavefront' $O/phases_at$at.txt | grep -E 'B jcalc|C kinematics|not stamped' | tr -s ' ' | tr '\n' '|')"
done
