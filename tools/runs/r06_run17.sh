#!/bin/bash
# round 6, run 17: hardware queue counts between 4 and 8, plain and collective path (same box)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run17; mkdir -p $out
A="--steps 40 --warmup 5 --no-cpu-baseline --no-host-frames"
for rep in 1 2; do for q in 4 5 6 8; do
GPU_MAX_HW_QUEUES=$q python bench.py $A 2>/dev/null | tail -1 > $out/plain_q${q}_$rep.json
GPU_MAX_HW_QUEUES=$q SLIDEO_BENCH_FORCE_DIST=1 SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py $A 2>/dev/null | tail -1 > $out/nccl_q${q}_$rep.json
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run17/*.json')):
    try: j=json.load(open(f)); print(f.split('/')[-1], j['value'], j['ms_per_step'])
    except Exception as e: print(f,'FAILED')
PY
