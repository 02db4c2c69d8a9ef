# Runs ON THE GPU BOX: quick look at a kernel change through the single-instantiation variant libraries
# (make VARIANT=_x1614 EXTRA=-DTDS_DEBUG_ONLY=1614 lib; ... _x3218): correctness of the two benchmark robots + speed.
export TMPDIR=/tmp
O=gpurun_out/${1:-r03v}; mkdir -p $O
P=$PWD/tiny-differentiable-simulator_amd
A="TDS_HIP_LIB=$P/libtds_hip_x1614.so"; Lk="TDS_HIP_LIB=$P/libtds_hip_x3218.so"
env $A timeout 600 python -m pytest tests/test_hip_parity.py tests/test_rings.py -m gpu -q --timeout 300 -k "(golden_single_steps and ant and not floating) or (every_ring_slot and ant-4096) or (full_size_closed_loop_every_env) or (stale and ant- and not floating)" > $O/pytest_ant.log 2>&1; tail -3 $O/pytest_ant.log
env $Lk timeout 600 python -m pytest tests/test_hip_parity.py tests/test_rings.py -m gpu -q --timeout 300 -k "(golden_single_steps and laikago and not floating) or (every_ring_slot and laikago_soft) or (stale and laikago and not floating)" > $O/pytest_laikago.log 2>&1; tail -3 $O/pytest_laikago.log
B="timeout 300 python bench.py --no-cpu-baseline --no-secondary"
env $A $B --steps 1000 --warmup 100 > $O/bench_1000.json 2> $O/bench_1000.err
env $A $B --steps 1000 --warmup 100 --records last > $O/bench_1000_last.json 2> $O/bench_1000_last.err
env $A TDS_HIP_LOOP_W2=0 $B --steps 1000 --warmup 100 > $O/bench_1000_w0.json 2> $O/bench_1000_w0.err
env $A $B --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
env $A $B --steps 1000 --warmup 100 --no-graph > $O/bench_1000_nograph.json 2> $O/bench_1000_nograph.err
env $A $B --steps 500 --warmup 50 --envs-per-gpu 8192 > $O/bench_8192.json 2> $O/bench_8192.err
env $A $B --steps 500 --warmup 50 --envs-per-gpu 16384 > $O/bench_16384.json 2> $O/bench_16384.err
env $Lk $B --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $O/bench_laikago.json 2> $O/bench_laikago.err
env $A timeout 200 python tools/profile_phases.py ant 4096 0 100 > $O/phases_ant4096.txt 2>&1
env $Lk timeout 200 python tools/profile_phases.py laikago_soft 8192 0 100 > $O/phases_laikago_soft8192.txt 2>&1
for f in $O/bench_*.json; do echo "$f: $(python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']), 'nonfinite=%d'%d['nonfinite_envs'])" 2>&1 | tail -1)"; done
sed -n 2,15p $O/phases_ant4096.txt | cut -c1-100; sed -n 20,44p $O/phases_ant4096.txt | cut -c1-100; sed -n 2,15p $O/phases_laikago_soft8192.txt
