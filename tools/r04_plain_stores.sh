#!/bin/bash
# Runs ON THE GPU BOX: record stores as ordinary write-back stores (slot 1, -DTDS_X_PLAIN_STORES) against streaming stores (the
# library): step time (same-process A/B) and HBM write traffic of a 1000-step and a 20-step ring launch.
export TMPDIR=/tmp
O=gpurun_out/r04q
P=gpurun_out/profiles
mkdir -p $O $P
timeout 200 python tools/ab_slots.py --slots 0,1 --reps 5 > $P/r04_ab_slots11_plain_stores.txt 2>&1; grep -v "max rel" $P/r04_ab_slots11_plain_stores.txt
for V in "library|" "plain|--option alt_build=1"; do
  IFS='|' read NAME ARGS <<< "$V"
  for K in "1000 100" "20 5"; do
    set -- $K
    i=0
    for CTRS in FETCH_SIZE WRITE_SIZE; do
      i=$((i+1))
      timeout 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/pmc_${NAME}_$1_$i -o p -- python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events $ARGS > $O/pmc_${NAME}_$1_$i.log 2>&1
    done
    echo "== $NAME, $1-step launch"; python tools/pmc_loop_summary.py $1 $O/pmc_${NAME}_$1_* | grep -v '^#' | cut -c1-130
    rm -rf $O/pmc_${NAME}_$1_*/
  done
done | tee -a $P/r04_ab_slots11_plain_stores.txt
