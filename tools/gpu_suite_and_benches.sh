#!/bin/bash
# Runs ON THE GPU BOX: the whole GPU test-suite, then the bench lines of the usual configs and the phase breakdown of
# the headline kernel -> gpurun_out/suite/ (what was run after every kernel change of round 2).
export TMPDIR=/tmp
O=gpurun_out/suite
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline"
$B --chains 1 > $O/bench_ant4096_c1.json 2>> $O/bench.err
$B > $O/bench_ant4096.json 2>> $O/bench.err
TDS_HIP_GRAM=1 $B > $O/bench_ant4096_gram.json 2>> $O/bench.err
$B --envs-per-gpu 8192 > $O/bench_ant8192.json 2>> $O/bench.err
$B --envs-per-gpu 16384 > $O/bench_ant16384.json 2>> $O/bench.err
$B --model pendulum5 --dtype f32 > $O/bench_pendulum5.json 2>> $O/bench.err
$B --model laikago_soft --envs-per-gpu 8192 > $O/bench_laikago_soft8192.json 2>> $O/bench.err
timeout 200 python tools/profile_phases.py ant 4096 0 100 > $O/phases_ant4096.txt 2>&1
timeout 200 python tools/profile_phases.py laikago_soft 8192 0 100 > $O/phases_laikago_soft8192.txt 2>&1
for f in $O/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.3f us'%(1000*d['ms_per_step']), d['config']['launch'][-60:])" 2>&1 | tail -1)"; done
head -16 $O/phases_ant4096.txt | tail -15
