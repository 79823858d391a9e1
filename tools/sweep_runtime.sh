#!/bin/bash
# usage (on the GPU box): tools/sweep_runtime.sh  -> A/B of runtime knobs (value, ms/step, kNN launch ms in the timed region, kNN alone)
run() { echo -n "$1: "; env $2 python bench.py --no-cpu-baseline $3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], (j.get('stage_ms_one_batch_in_flight') or {}).get('knn'))"; }
for w in cfg1 cfg4 tiny; do
run "$w auto(mfma2)" "A=1" "--workload $w --steps 20 --warmup 4"
run "$w mfma4" "A=1" "--workload $w --steps 20 --warmup 4 --knn mfma4"
done
run "headline auto" "A=1" "--steps 30 --warmup 5"
