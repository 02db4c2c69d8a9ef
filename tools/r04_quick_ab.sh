#!/bin/bash
# Runs ON THE GPU BOX: same-process A/B of whatever experiment slots the library carries (tools/build_alt.sh).
export TMPDIR=/tmp
P=gpurun_out/profiles
mkdir -p $P
timeout 250 python tools/ab_slots.py --slots ${SLOTS:-0,1,2,3,4,5,6} --reps ${REPS:-5} $ARGS > $P/${OUT:-r04_ab_slots_quick}.txt 2>&1; cat $P/${OUT:-r04_ab_slots_quick}.txt
if [ -n "$ARGS2" ]; then timeout 250 python tools/ab_slots.py --slots ${SLOTS:-0,1,2,3,4,5,6} --reps ${REPS:-5} $ARGS2 > $P/${OUT:-r04_ab_slots_quick}_2.txt 2>&1; cat $P/${OUT:-r04_ab_slots_quick}_2.txt; fi
