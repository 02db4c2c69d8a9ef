export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
timeout 900 python -m pytest tests/test_rings.py tests/test_multi_gpu.py tests/test_shard_two_ranks_one_gpu.py -m gpu -q --timeout 300 > $O/pytest_a.log 2>&1; tail -5 $O/pytest_a.log
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout 300 -k "step_many or stale or substeps or auto_reset" > $O/pytest_b.log 2>&1; tail -4 $O/pytest_b.log
B="timeout 300 python bench.py --no-cpu-baseline"
for W in 1 0; do
TDS_HIP_LOOP_W2=$W $B --steps 20 --warmup 5 --no-secondary > $O/bench_20_w$W.json 2> $O/bench_20_w$W.err
TDS_HIP_LOOP_W2=$W $B --steps 1000 --warmup 100 --no-secondary > $O/bench_1000_w$W.json 2> $O/bench_1000_w$W.err
TDS_HIP_LOOP_W2=$W $B --steps 1000 --warmup 100 --records last --no-secondary > $O/bench_1000_last_w$W.json 2> $O/bench_1000_last_w$W.err
TDS_HIP_LOOP_W2=$W $B --steps 1000 --warmup 100 --force-gather > $O/bench_fg1000_w$W.json 2> $O/bench_fg1000_w$W.err
done
for f in $O/bench_*.json; do echo "$f: $(python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']), d['config'].get('exchange_form'), 'nonfinite=%d'%d['nonfinite_envs'])
except Exception as e:
    print('ERR', e)
P
)"; done
tail -2 $O/*.err | cut -c1-200 | grep -v "^$" | head -20
