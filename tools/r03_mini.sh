export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; tail -6 $O/pytest.log
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -i -E "icache|IFETCH|SQC_" $O/avail.txt | cut -c1-140 | sort -u | head -60
for FORM in loop w2; do
  EX=$( [ $FORM = w2 ] && echo --no-graph )
  i=0
  for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
              "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
              "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/sq_${FORM}_$i -o p -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events $EX > $O/sq_${FORM}_$i.log 2>&1
  done
  if [ $FORM = loop ]; then python tools/pmc_loop_summary.py 200 $O/sq_${FORM}_* > $O/sq_counters_$FORM.txt 2>&1
  else python tools/pmc_summary.py $O/sq_${FORM}_* > $O/sq_counters_$FORM.txt 2>&1; fi
  rm -rf $O/sq_${FORM}_?
  cat $O/sq_counters_$FORM.txt | cut -c1-160
done
