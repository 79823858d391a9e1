#!/bin/bash
# A/B over (build flags, environment, bench arguments): tools/ab_matrix.sh <out name> 'name|-D flags|ENV=.. ENV2=..|bench args' ...
# (tools/ab_variants.sh with the other two axes).  REPS rounds, interleaved; results under gpurun_out/abm_<out name>/.
cd $GRAFT_REPO_ROOT
out=gpurun_out/abm_$1; shift; mkdir -p $out
declare -A built
specs=("$@")
for s in "${specs[@]}"; do
  IFS='|' read -r n f e a <<< "$s"
  tag=$(echo "x$f" | md5sum | cut -c1-8)
  if [ -z "${built[$tag]}" ]; then
    SLIDEO_HIP_EXTRA_FLAGS="$f" python -m slideo_amd.build --tag $tag > $out/build_$tag.log 2>&1 || { echo "build of $n failed"; tail -5 $out/build_$tag.log; }
    built[$tag]=1
  fi
done
for rep in $(seq 1 ${REPS:-2}); do for s in "${specs[@]}"; do
  IFS='|' read -r n f e a <<< "$s"
  tag=$(echo "x$f" | md5sum | cut -c1-8)
  env $e SLIDEO_LIB_PATH=slideo_amd/lib/variants/$tag/libslideo_amd.so timeout 300 python bench.py ${a:---steps 40 --warmup 5} --no-cpu-baseline 2>$out/${n}_$rep.err | tail -1 > $out/${n}_$rep.json
done; done
python - "$out" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try: j=json.load(open(f))
    except Exception as e: print(os.path.basename(f), 'FAILED', open(f.replace('.json','.err')).read()[-300:]); continue
    r=j.get('roofline',{}); a=j.get('stage_ms_one_batch_in_flight') or j.get('stage_ms_per_batch')
    print('%-24s %9.1f f/s %7.3f ms/step  knn %.2f  alone %s' % (os.path.basename(f)[:-5], j['value'], j['ms_per_step'], r.get('avg_launch_ms',0), ' '.join('%s %.2f'%(k,v) for k,v in a.items()) if isinstance(a,dict) else a))
PY
