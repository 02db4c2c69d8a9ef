export TMPDIR=/tmp
O=gpurun_out/r03j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; tail -6 $O/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-secondary"
$B --steps 1000 --warmup 100 > $O/bench_1000.json 2> $O/bench_1000.err
TDS_HIP_DENSE=1 $B --steps 1000 --warmup 100 > $O/bench_1000_dense.json 2> $O/bench_1000_dense.err
$B --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
$B --steps 1000 --warmup 100 --no-graph > $O/bench_1000_nograph.json 2> $O/bench_1000_nograph.err
$B --steps 500 --warmup 50 --envs-per-gpu 8192 > $O/bench_8192.json 2> $O/bench_8192.err
$B --steps 500 --warmup 50 --envs-per-gpu 16384 > $O/bench_16384.json 2> $O/bench_16384.err
$B --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $O/bench_laikago.json 2> $O/bench_laikago.err
TDS_HIP_DENSE=1 $B --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $O/bench_laikago_dense.json 2> $O/bench_laikago_dense.err
timeout 200 python tools/profile_phases.py ant 4096 0 100 > $O/phases_ant4096.txt 2>&1
timeout 200 python tools/profile_phases.py laikago_soft 8192 0 100 > $O/phases_laikago_soft8192.txt 2>&1
for f in $O/bench_*.json; do echo "$f: $(python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']), 'nonfinite=%d'%d['nonfinite_envs'])" 2>&1 | tail -1)"; done
head -50 $O/phases_ant4096.txt | cut -c1-120; head -16 $O/phases_laikago_soft8192.txt
