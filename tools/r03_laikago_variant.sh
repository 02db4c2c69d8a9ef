# Runs ON THE GPU BOX: the Laikago (32 lanes, 18 dof) kernels of a variant library against the default build —
# parity of the Laikago models, config 4 through the chained graphs and (TDS_HIP_STEP_MANY_LOOP=1) as one step-loop launch
export TMPDIR=/tmp
V=${1:-x3218b}; O=gpurun_out/lk_$V; mkdir -p $O
P=$PWD/tiny-differentiable-simulator_amd
Lk="TDS_HIP_LIB=$P/libtds_hip_$V.so"
env $Lk timeout 600 python -m pytest tests/test_hip_parity.py tests/test_rings.py -m gpu -q --timeout 300 -k "(golden_single_steps and laikago and not floating) or (every_ring_slot and laikago_soft) or (stale and laikago and not floating) or (substeps and laikago-)" > $O/pytest_laikago.log 2>&1; tail -3 $O/pytest_laikago.log
B="timeout 300 python bench.py --no-cpu-baseline --no-secondary --model laikago_soft --envs-per-gpu 8192 --steps 500 --warmup 50"
$B > $O/bench_default_graphs.json 2> $O/e1
env $Lk $B > $O/bench_variant_graphs.json 2> $O/e2
env TDS_HIP_STEP_MANY_LOOP=1 $B > $O/bench_default_loop.json 2> $O/e3
env $Lk TDS_HIP_STEP_MANY_LOOP=1 $B > $O/bench_variant_loop.json 2> $O/e4
env $Lk TDS_HIP_STEP_MANY_LOOP=1 $B --records last > $O/bench_variant_loop_last.json 2> $O/e5
B2="timeout 300 python bench.py --no-cpu-baseline --no-secondary --model laikago_soft --envs-per-gpu 2048 --steps 500 --warmup 50"
$B2 > $O/bench2048_default.json 2> $O/e6
env $Lk $B2 > $O/bench2048_variant.json 2> $O/e7
env $Lk TDS_HIP_STEP_MANY_LOOP=1 $B2 > $O/bench2048_variant_loop.json 2> $O/e8
for f in $O/bench*.json; do echo "$f: $(python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']), 'nonfinite=%s'%d['config'].get('nonfinite_envs'), d['config'].get('launch','')[:60])" 2>&1 | tail -1)"; done
env $Lk timeout 200 python tools/profile_phases.py laikago_soft 8192 0 100 2>/dev/null | sed -n 1,16p
