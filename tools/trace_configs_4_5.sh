# Runs ON THE GPU BOX: rocprofv3 kernel trace + stats of configs 4 and 5 (bench.py --steps 500) -> gpurun_out/profiles/<tag>_*_kernel_stats.txt
# usage: TAG=r06e bash tools/trace_configs_4_5.sh
TAG=${TAG:-r06e}
export TMPDIR=/tmp
O=gpurun_out/work_${TAG}; P=gpurun_out/profiles; mkdir -p $O $P
for W in "laikago_soft 8192 f64" "ant 8192 f64"; do set -- $W
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$1 -o k -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-secondary --model $1 --envs-per-gpu $2 > $O/kt_$1.log 2>&1
  DB=$(ls $O/kt_$1/*.db $O/kt_$1/*/*.db 2>/dev/null | head -1)
  python tools/rocprof_summary.py "$DB" > $P/${TAG}_${1}${2}_f64_kernel_stats.txt 2>&1
  grep -h '"value"' $O/kt_$1.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('# bench line of the same command: value %.4g, %.2f us/step, roofline.kernel_ms_avg %.4f (steps_per_launch %s)'%(d['value'],1e3*d['ms_per_step'],d['roofline']['kernel_ms_avg'],d['roofline'].get('steps_per_launch')))" >> $P/${TAG}_${1}${2}_f64_kernel_stats.txt
  head -5 $P/${TAG}_${1}${2}_f64_kernel_stats.txt | cut -c1-170; tail -1 $P/${TAG}_${1}${2}_f64_kernel_stats.txt
  rm -rf $O/kt_$1
done
