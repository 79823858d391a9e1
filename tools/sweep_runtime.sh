#!/bin/bash
# usage (on the GPU box): tools/sweep_runtime.sh  -> A/B of runtime knobs on the headline workload (value, ms/step, kNN launch ms in the timed region, kNN alone)
run() { echo -n "$1: "; env $2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], (j.get('stage_ms_one_batch_in_flight') or {}).get('knn'))"; }
for i in 1 2; do
run "mfma2 (default)" "A=1" ""
run "mfma4" "A=1" "--knn mfma4"
done
run "mfma2 inflight3" "A=1" "--inflight 3"
run "mfma2 inflight2" "A=1" "--inflight 2"
run "mfma2 chain0" "SLIDEO_ORB_CHAIN=0" ""
