#!/bin/bash
# round 6, run 10: the two repaired bench-contract tests; the collective path at world size 1 against the plain command at FULL size (headline and cfg3)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run10; mkdir -p $out
timeout 1500 python -m pytest tests/test_bench_contract.py -q -m gpu -k "single_gpu_line or eight_ranks" > $out/tests.log 2>&1; tail -5 $out/tests.log
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1 > $out/plain_$rep.json
SLIDEO_BENCH_FORCE_DIST=1 SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1 > $out/nccl1_$rep.json
done
python bench.py --workload cfg3 --steps 20 --warmup 2 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1 > $out/cfg3_plain.json
SLIDEO_BENCH_FORCE_DIST=1 SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py --workload cfg3 --steps 20 --warmup 2 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1 > $out/cfg3_nccl1.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run10/*.json')):
    try: j=json.load(open(f)); print(f.split('/')[-1], j['value'], j['ms_per_step'], j['config'].get('collective'))
    except Exception as e: print(f, 'FAILED', e)
PY
