# Runs ON THE GPU BOX: the segmented leg scan (DevModel::leg_len) against the level loop (TDS_HIP_NO_LEGSCAN=1), same library
export TMPDIR=/tmp
O=gpurun_out/legscan; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_rings.py -m gpu -q --timeout 600 -k "(golden_single_steps and (ant or laikago) and not floating) or every_ring_slot or full_size_closed_loop_every_env or (stale and (ant- or laikago) and not floating) or (substeps and (ant or laikago-))" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-secondary"
for V in scan levels; do
  E=""; [ $V = levels ] && E="TDS_HIP_NO_LEGSCAN=1"
  env $E $B --steps 1000 --warmup 100 > $O/ant4096_1000_$V.json 2>/dev/null
  env $E $B --steps 20 --warmup 5 > $O/ant4096_20_$V.json 2>/dev/null
  env $E $B --steps 1000 --warmup 100 --no-graph > $O/ant4096_nograph_$V.json 2>/dev/null
  env $E $B --steps 500 --warmup 50 --envs-per-gpu 8192 > $O/ant8192_$V.json 2>/dev/null
  env $E $B --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $O/laikago8192_$V.json 2>/dev/null
done
for f in $O/*.json; do echo "$f: $(python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']))" 2>&1 | tail -1)"; done
timeout 200 python tools/profile_phases.py ant 4096 0 100 2>/dev/null > $O/phases_ant.txt; sed -n 2,5p $O/phases_ant.txt | cut -c1-100; sed -n 20,34p $O/phases_ant.txt | cut -c1-100
timeout 200 python tools/profile_phases.py laikago_soft 8192 0 100 2>/dev/null > $O/phases_laikago.txt; sed -n 1,6p $O/phases_laikago.txt | cut -c1-100
