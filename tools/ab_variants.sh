#!/bin/bash
# A/B of kernel variants on the GPU box: tools/ab_variants.sh "<bench args>" name1:"-DFLAG=1 ..." name2:"..." ...
# Every variant is the product source built with extra -D flags into slideo_amd/lib/variants/<name>/ (base = no flags); the
# bench runs interleaved, REPS rounds (default 2), and a table is printed.  Results under gpurun_out/ab_<first name>/.
cd $GRAFT_REPO_ROOT
args="$1"; shift
out=gpurun_out/ab_$(echo "$1" | cut -d: -f1); mkdir -p $out
names=(base)
SLIDEO_HIP_EXTRA_FLAGS="" python -m slideo_amd.build --tag base > /dev/null || exit 1
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  SLIDEO_HIP_EXTRA_FLAGS="$f" python -m slideo_amd.build --tag $n > $out/build_$n.log 2>&1 || { echo "build of $n failed"; tail -5 $out/build_$n.log; continue; }
  names+=($n)
done
for rep in $(seq 1 ${REPS:-2}); do for n in "${names[@]}"; do
  SLIDEO_LIB_PATH=slideo_amd/lib/variants/$n/libslideo_amd.so python bench.py $args --no-cpu-baseline 2>$out/${n}_$rep.err | tail -1 > $out/${n}_$rep.json
done; done
python - "$out" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try: j=json.load(open(f))
    except Exception as e: print(os.path.basename(f), 'FAILED', open(f.replace('.json','.err')).read()[-300:]); continue
    r=j.get('roofline',{}); a=j.get('stage_ms_one_batch_in_flight') or j.get('stage_ms_per_batch')
    print('%-28s %9.1f f/s %7.3f ms/step  knn %.2f  alone %s  acc %s' % (os.path.basename(f), j['value'], j['ms_per_step'], r.get('avg_launch_ms',0), a, j['config'].get('accuracy_vs_synthetic_truth')))
PY
