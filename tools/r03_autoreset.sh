export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_rings.py -m gpu -q --timeout 300 -k "auto_reset or pool" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python tools/auto_reset_modes.py > $O/auto_reset_modes.txt 2>&1
timeout 900 python tools/auto_reset_modes.py many >> $O/auto_reset_modes.txt 2>&1
TDS_HIP_LOOP_W2=2 timeout 900 python tools/auto_reset_modes.py many > $O/auto_reset_modes_w2nopool.txt 2>&1
cat $O/auto_reset_modes.txt | cut -c1-250; echo ==== ; cat $O/auto_reset_modes_w2nopool.txt | cut -c1-250
B="timeout 300 python bench.py --no-cpu-baseline"
TDS_BENCH_RCCL_SINGLE=0 $B --steps 1000 --warmup 100 --force-gather > $O/bench_fg1000_copy.json 2> $O/bench_fg1000_copy.err
$B --steps 1000 --warmup 100 --force-gather > $O/bench_fg1000_rccl.json 2> $O/bench_fg1000_rccl.err
TDS_HIP_LOOP_W2=0 TDS_BENCH_RCCL_SINGLE=0 $B --steps 1000 --warmup 100 --force-gather > $O/bench_fg1000_copy_w0.json 2> $O/bench_fg1000_copy_w0.err
for f in $O/bench_*.json; do echo "$f: $(python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']))" 2>&1 | tail -1)"; done
