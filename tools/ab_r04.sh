cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for p in 0 1 2; do
  SLIDEO_VERIFY_PRIO=$p python bench.py --steps 60 --warmup 8 --no-cpu-baseline 2>/tmp/o.err | tail -1 > /tmp/o.json; [ -s /tmp/o.json ] || tail -3 /tmp/o.err
  python - $p <<'PY'
import json,sys
try:
    j=json.load(open('/tmp/o.json')); r=j['roofline']
except Exception:
    print(sys.argv[1], 'FAILED'); sys.exit(0)
print('vprio', sys.argv[1], j['value'], j['ms_per_step'], 'knn', r['avg_launch_ms'], 'frac', r['frac'], j['stage_ms_per_step'])
PY
done; done
