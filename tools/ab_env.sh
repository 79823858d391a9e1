#!/bin/bash
# A/B of ENVIRONMENT settings (and / or libraries) on one GPU box, the bench interleaved over REPS rounds (default 3):
#   tools/ab_env.sh <out name> "<bench args>" name1="ENV1=a ENV2=b" name2="" name3="SLIDEO_LIB_PATH=path/to/lib.so" ...
# results under gpurun_out/abe_<out name>/; one summary line per name.
cd $GRAFT_REPO_ROOT
out=gpurun_out/abe_$1; shift; args="$1"; shift
mkdir -p $out
for rep in $(seq 1 ${REPS:-3}); do for v in "$@"; do
  n=${v%%=*}; e=${v#*=}
  env $e python bench.py $args --no-cpu-baseline 2>$out/${n}_$rep.err | tail -1 > $out/${n}_$rep.json
done; done
python - "$out" <<'PY'
import json,glob,sys,os,collections
rows=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try: j=json.load(open(f))
    except Exception as e: print(os.path.basename(f), 'FAILED', open(f.replace('.json','.err')).read()[-400:]); continue
    a=j.get('stage_ms_one_batch_in_flight') or {}
    r=j.get('roofline') or {}
    rows[os.path.basename(f).rsplit('_',1)[0]].append((j['ms_per_step'], j['value'], a, r.get('avg_launch_ms'), j['config'].get('accuracy_vs_synthetic_truth')))
for n,v in rows.items():
    print('%-14s ms/step %s  frames/s %s  launch_ms %s  acc %s  alone(last) %s' % (n, ' '.join('%.3f'%x[0] for x in v), ' '.join('%.0f'%x[1] for x in v), ' '.join('%.2f'%(x[3] or 0) for x in v), v[-1][4], ' '.join('%s %.2f'%(k,q) for k,q in v[-1][2].items())))
PY
