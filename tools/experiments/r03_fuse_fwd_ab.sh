# Runs ON THE GPU BOX: forward substitution of the forward dynamics fused into the LDL^T (f1) against the loop behind it (f0);
# Ant-only variant libraries (make VARIANT=_x1614f1 EXTRA="-DTDS_DEBUG_ONLY=1614 -DTDS_FUSE_FWD=1" lib)
export TMPDIR=/tmp
O=gpurun_out/fusefwd; mkdir -p $O
P=$PWD/tiny-differentiable-simulator_amd
A="TDS_HIP_LIB=$P/libtds_hip_x1614f1.so"
env $A timeout 600 python -m pytest tests/test_rings.py tests/test_hip_parity.py -m gpu -q --timeout 300 -k "(every_ring_slot and ant-4096) or full_size_closed_loop_every_env or (ring and ant)" > $O/pytest_ant.log 2>&1; tail -1 $O/pytest_ant.log
B="timeout 300 python bench.py --no-cpu-baseline --no-secondary"
for rep in 1 2; do for v in f1 f0; do
A="TDS_HIP_LIB=$P/libtds_hip_x1614$v.so"
env $A $B --steps 1000 --warmup 100 > $O/ant4096_1000_${v}_$rep.json 2>/dev/null
env $A $B --steps 20 --warmup 5 > $O/ant4096_20_${v}_$rep.json 2>/dev/null
env $A $B --steps 1000 --warmup 100 --records last > $O/ant4096_last_${v}_$rep.json 2>/dev/null
done; done
for f in $O/*.json; do echo "$(basename $f): $(python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']))")"; done
