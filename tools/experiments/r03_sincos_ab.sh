# Runs ON THE GPU BOX: jcalc's sin / cos by sincos_joint (s1) against the library's sincos (s0), single-instantiation
# variant libraries (make VARIANT=_x1614s0 EXTRA="-DTDS_DEBUG_ONLY=1614 -DTDS_FAST_SINCOS=0" lib; ...)
export TMPDIR=/tmp
O=gpurun_out/sincos; mkdir -p $O
P=$PWD/tiny-differentiable-simulator_amd
for v in s1 s0; do
A="TDS_HIP_LIB=$P/libtds_hip_x1614$v.so"; Lk="TDS_HIP_LIB=$P/libtds_hip_x3218$v.so"
if [ $v = s1 ]; then
env $A timeout 600 python -m pytest tests/test_hip_parity.py tests/test_rings.py -m gpu -q --timeout 300 -k "(golden_single_steps and ant and not floating) or (every_ring_slot and ant-4096) or (full_size_closed_loop_every_env) or (stale and ant- and not floating)" > $O/pytest_ant.log 2>&1; tail -1 $O/pytest_ant.log
env $Lk timeout 600 python -m pytest tests/test_hip_parity.py tests/test_rings.py -m gpu -q --timeout 300 -k "(golden_single_steps and laikago and not floating) or (every_ring_slot and laikago_soft) or (stale and laikago and not floating)" > $O/pytest_laikago.log 2>&1; tail -1 $O/pytest_laikago.log
fi
B="timeout 300 python bench.py --no-cpu-baseline --no-secondary"
env $A $B --steps 1000 --warmup 100 > $O/ant4096_1000_$v.json 2>/dev/null
env $A $B --steps 1000 --warmup 100 --no-graph > $O/ant4096_nograph_$v.json 2>/dev/null
env $A $B --steps 500 --warmup 50 --envs-per-gpu 8192 > $O/ant8192_$v.json 2>/dev/null
env $Lk $B --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $O/laikago8192_$v.json 2>/dev/null
done
for f in $O/*.json; do echo "$(basename $f): $(python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']))")"; done
