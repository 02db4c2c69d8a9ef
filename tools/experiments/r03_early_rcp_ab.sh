# Runs ON THE GPU BOX: LDL^T with the next pivot's reciprocal one column early (r1) against the plain form (r0); Ant-only
# variant libraries (make VARIANT=_x1614r0 EXTRA="-DTDS_DEBUG_ONLY=1614 -DTDS_H_EARLY_RCP=0" lib; ... r1)
export TMPDIR=/tmp
O=gpurun_out/earlyrcp; mkdir -p $O
P=$PWD/tiny-differentiable-simulator_amd
A="TDS_HIP_LIB=$P/libtds_hip_x1614r1.so"
env $A timeout 600 python -m pytest tests/test_rings.py tests/test_hip_parity.py -m gpu -q --timeout 300 -k "(every_ring_slot and ant-4096) or full_size_closed_loop_every_env" > $O/pytest_ant.log 2>&1; tail -1 $O/pytest_ant.log
B="timeout 300 python bench.py --no-cpu-baseline --no-secondary"
for rep in 1 2; do for v in r1 r0; do
A="TDS_HIP_LIB=$P/libtds_hip_x1614$v.so"
env $A $B --steps 1000 --warmup 100 > $O/ant4096_1000_${v}_$rep.json 2>/dev/null
env $A $B --steps 1000 --warmup 100 --no-graph > $O/ant4096_nograph_${v}_$rep.json 2>/dev/null
env $A $B --steps 500 --warmup 50 --envs-per-gpu 8192 > $O/ant8192_${v}_$rep.json 2>/dev/null
done; done
for f in $O/*.json; do echo "$(basename $f): $(python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']))")"; done
env TDS_HIP_LIB=$P/libtds_hip_x1614r1.so timeout 200 python tools/profile_phases.py ant 4096 0 100 2>/dev/null | sed -n 2,15p
