cd $GRAFT_REPO_ROOT
SLIDEO_HIP_EXTRA_FLAGS="" python -m slideo_amd.build --tag s4 > /dev/null
SLIDEO_HIP_EXTRA_FLAGS="-DSLIDEO_NSLOTS=6" python -m slideo_amd.build --tag s6 > /dev/null
SLIDEO_HIP_EXTRA_FLAGS="-DSLIDEO_NSLOTS=8" python -m slideo_amd.build --tag s8 > /dev/null
for rep in 1 2; do for v in s4:0 s4:1 s6:0 s6:1 s8:0 s8:1; do
  t=${v%%:*}; p=${v#*:}
  SLIDEO_KNN_PRIO=$p SLIDEO_LIB_PATH=slideo_amd/lib/variants/$t/libslideo_amd.so python bench.py --steps 60 --warmup 8 --no-cpu-baseline 2>/tmp/o.err | tail -1 > /tmp/o.json; [ -s /tmp/o.json ] || tail -3 /tmp/o.err
  python - $t $p <<'PY'
import json,sys
try:
    j=json.load(open('/tmp/o.json')); r=j['roofline']
except Exception:
    print(sys.argv[1], sys.argv[2], 'FAILED'); sys.exit(0)
print(sys.argv[1], 'prio', sys.argv[2], j['value'], j['ms_per_step'], 'knn', r['avg_launch_ms'], 'frac', r['frac'], j['config']['parallelism'][-40:])
PY
done; done
