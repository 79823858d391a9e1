#!/bin/bash
# round 5, GPU run 1: baseline bench under the ABI-6 code, SLIDEO_CU_SPLIT sweep, variant sensitivity, group rate, the whole GPU suite
O=gpurun_out/r05a; mkdir -p $O
python bench.py --steps 100 --warmup 5 > $O/bench_headline.json 2> $O/bench_headline.err
for N in 160 176 192 208 224; do
  SLIDEO_CU_SPLIT=$N timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_split_$N.json 2> $O/bench_split_$N.err
done
for N in 192 224; do
  SLIDEO_CU_SPLIT=$N SLIDEO_CU_SPLIT_OTHERS=0 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_split_${N}_others_anywhere.json 2> $O/bench_split_${N}_oa.err
done
SLIDEO_CU_SPLIT=192 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-overlap > $O/bench_split_192_no_overlap.json 2>&1
timeout 900 python tools/variant_sensitivity.py > $O/variant_sensitivity.json 2> $O/variant_sensitivity.err
timeout 900 python tools/group_rate.py > $O/group_rate.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $O/tests_full.log 2>&1
tail -5 $O/tests_full.log
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][0]
    r=j.get('roofline',{})
    print(j['value'], j['ms_per_step'], r.get('avg_launch_ms'), j.get('stage_ms_per_step'), j.get('stage_ms_one_batch_in_flight'))
except Exception as e: print('ERR', e)
PY
done
