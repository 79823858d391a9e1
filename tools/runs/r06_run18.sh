#!/bin/bash
# round 6, run 18: slot streams picked by measurement (SLIDEO_STREAM_PICK): the collective path with the DEFAULT four hardware queues; group / parity tests
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run18; mkdir -p $out
A="--steps 40 --warmup 5 --no-cpu-baseline --no-host-frames"
for rep in 1 2; do
python bench.py $A 2>/dev/null | tail -1 > $out/plain_pick_$rep.json
SLIDEO_STREAM_PICK=0 python bench.py $A 2>/dev/null | tail -1 > $out/plain_nopick_$rep.json
GPU_MAX_HW_QUEUES=4 SLIDEO_BENCH_FORCE_DIST=1 SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py $A 2>/dev/null | tail -1 > $out/nccl_q4_pick_$rep.json
GPU_MAX_HW_QUEUES=4 SLIDEO_STREAM_PICK=0 SLIDEO_BENCH_FORCE_DIST=1 SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py $A 2>/dev/null | tail -1 > $out/nccl_q4_nopick_$rep.json
GPU_MAX_HW_QUEUES=8 SLIDEO_BENCH_FORCE_DIST=1 SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py $A 2>/dev/null | tail -1 > $out/nccl_q8_pick_$rep.json
GPU_MAX_HW_QUEUES=2 python bench.py $A 2>/dev/null | tail -1 > $out/plain_q2_pick_$rep.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run18/*.json')):
    try: j=json.load(open(f)); print(f.split('/')[-1], j['value'], j['ms_per_step'], 'db build s', j['config']['page_db_build_s'])
    except Exception as e: print(f,'FAILED')
PY
timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_matching_mirror.py tests/test_capi_load.py -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $out/parity.log 2>&1; tail -3 $out/parity.log
