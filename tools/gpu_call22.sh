#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/c22
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_f32.py -m gpu -q -k "ant or golden or full_size or equivar or permut" > $O/pytest.log 2>&1; tail -8 $O/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline"
$B --chains 1 > $O/bench_ant4096_c1.json 2>> $O/bench.err
$B > $O/bench_ant4096.json 2>> $O/bench.err
for at in 1 2 3; do
TDS_GRAM_STAMP_AT=$at timeout 200 python tools/profile_phases.py ant 4096 0 100 > $O/phases_at$at.txt 2>&1
echo "at=$at"; grep -A16 "  main wavefront" $O/phases_at$at.txt | sed -n 10,16p
done
for f in $O/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.3f us'%(1000*d['ms_per_step']), d['config']['launch'][-40:])" 2>&1 | tail -1)"; done
