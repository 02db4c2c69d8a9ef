export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
for R in 1 0; do
rocprofv3 --kernel-trace --stats -d $O/kt_$R -o k -- python tools/trace_ring_exchange.py $R 4096 3 > $O/trace_$R.log 2>&1
DB=$(ls $O/kt_$R/*.db $O/kt_$R/*/*.db 2>/dev/null | head -1)
python tools/rocprof_all_dispatches.py "$DB" -150 150 > $O/dispatches_rccl$R.txt 2>&1
python tools/rocprof_summary.py "$DB" | head -12 > $O/stats_rccl$R.txt
rm -rf $O/kt_$R
tail -2 $O/trace_$R.log
done
head -70 $O/dispatches_rccl0.txt | cut -c1-140
