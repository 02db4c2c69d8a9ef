export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_quad.py tests/test_rings.py -x -q -m gpu 2>&1 | tail -5
for N in 8192 7168 6144 4096; do
for w in 0 1 2; do
for ar in "--no-auto-reset" ""; do
  timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu $N --option quad_wide=$w $ar > /tmp/q.json 2>/tmp/q.err
  python3 -c "
import json;d=json.load(open('/tmp/q.json'));print($N,'quad_wide=$w','$ar','value %.4g us/step %.2f'%(d['value'],1e3*d['ms_per_step']))" || tail -3 /tmp/q.err
done; done; done
