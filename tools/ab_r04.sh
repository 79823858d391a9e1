cd $GRAFT_REPO_ROOT
for rep in 1 2; do for c in 1 0; do
  SLIDEO_ORB_CHAIN=$c python bench.py --steps 60 --warmup 8 --no-cpu-baseline 2>/tmp/o.err | tail -1 > /tmp/o.json; [ -s /tmp/o.json ] || tail -3 /tmp/o.err
  python - $c <<'PY'
import json,sys
j=json.load(open('/tmp/o.json')); r=j['roofline']
print('orb_chain', sys.argv[1], j['value'], j['ms_per_step'], 'knn', r['avg_launch_ms'], j['stage_ms_per_step'])
PY
done; done
for i in 2 3 4; do python bench.py --steps 60 --warmup 8 --no-cpu-baseline --inflight $i 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.readline()); print('inflight $i', j['value'], j['ms_per_step'], j['stage_ms_per_step'])"; done
