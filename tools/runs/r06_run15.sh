#!/bin/bash
# round 6, run 15: the collective path with its hardware queues (bench.py sets GPU_MAX_HW_QUEUES=8): world size 1 nccl against plain; the repaired contract tests
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run15; mkdir -p $out
A="--steps 20 --warmup 5 --no-cpu-baseline --no-host-frames"
for rep in 1 2; do
python bench.py $A 2>/dev/null | tail -1 > $out/plain_$rep.json
SLIDEO_BENCH_FORCE_DIST=1 SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py $A 2>/dev/null | tail -1 > $out/nccl1_$rep.json
done
python bench.py --workload cfg3 --steps 20 --warmup 2 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1 > $out/cfg3_plain.json
SLIDEO_BENCH_FORCE_DIST=1 SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py --workload cfg3 --steps 20 --warmup 2 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1 > $out/cfg3_nccl1.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run15/*.json')):
    try: j=json.load(open(f)); print(f.split('/')[-1], j['value'], j['ms_per_step'], j['config'].get('collective'), j['config'].get('gpu_max_hw_queues'))
    except Exception as e: print(f, 'FAILED', e)
PY
timeout 2400 python -m pytest tests/test_bench_contract.py tests/test_bench_launch.py -q -m gpu > $out/tests.log 2>&1; tail -5 $out/tests.log
