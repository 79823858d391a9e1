#!/bin/bash
# A/B of whole LIBRARIES on one GPU box (box-to-box spread is +-2 %, a kernel change is often less):
#   tools/ab_libs.sh <out name> "<bench args>" name1=path/to/libslideo_amd.so name2=... ; "product" = the in-tree build.
# The bench runs interleaved, REPS rounds (default 3); results under gpurun_out/abl_<out name>/.
cd $GRAFT_REPO_ROOT
out=gpurun_out/abl_$1; shift; args="$1"; shift
mkdir -p $out
for rep in $(seq 1 ${REPS:-3}); do for v in "$@"; do
  n=${v%%=*}; p=${v#*=}
  if [ "$p" = product ]; then env -u SLIDEO_LIB_PATH python bench.py $args --no-cpu-baseline 2>$out/${n}_$rep.err | tail -1 > $out/${n}_$rep.json
  else SLIDEO_LIB_PATH=$p python bench.py $args --no-cpu-baseline 2>$out/${n}_$rep.err | tail -1 > $out/${n}_$rep.json; fi
done; done
python - "$out" <<'PY'
import json,glob,sys,os,collections
rows=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try: j=json.load(open(f))
    except Exception as e: print(os.path.basename(f), 'FAILED', open(f.replace('.json','.err')).read()[-300:]); continue
    a=j.get('stage_ms_one_batch_in_flight') or {}
    rows[os.path.basename(f).rsplit('_',1)[0]].append((j['ms_per_step'], j['value'], a))
for n,v in rows.items():
    print('%-16s ms/step %s   frames/s %s   alone(last) %s' % (n, ' '.join('%.3f'%x[0] for x in v), ' '.join('%.0f'%x[1] for x in v), ' '.join('%s %.2f'%(k,q) for k,q in v[-1][2].items())))
PY
