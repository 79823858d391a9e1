#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run13; mkdir -p $out
A="--steps 20 --warmup 5 --no-cpu-baseline --no-host-frames"
for rep in 1 2; do
python bench.py $A 2>/dev/null | tail -1 > $out/a_plain_$rep.json
SLIDEO_BENCH_FORCE_DIST=1 SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py $A 2>/dev/null | tail -1 > $out/c_nccl_$rep.json
SLIDEO_BENCH_SKIP_GATHER=1 SLIDEO_BENCH_FORCE_DIST=1 SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py $A 2>/dev/null | tail -1 > $out/e_nccl_init_only_$rep.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run13/*.json')):
    j=json.load(open(f)); print(f.split('/')[-1], j['value'], j['ms_per_step'], j['config'].get('host_ms_per_step'))
PY
