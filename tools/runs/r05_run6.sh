#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
timeout 600 python bench.py --steps 60 --warmup 6 --no-cpu-baseline > gpurun_out/r05f/u192.json 2> gpurun_out/r05f/u192.err
timeout 600 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --unit 256 > gpurun_out/r05f/u256.json 2> gpurun_out/r05f/u256.err
timeout 600 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --unit 176 > gpurun_out/r05f/u176.json 2> gpurun_out/r05f/u176.err
timeout 600 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --unit 200 > gpurun_out/r05f/u200.json 2> gpurun_out/r05f/u200.err
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05f/u192_driver_form.json 2> gpurun_out/r05f/u192_driver_form.err
timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -q -x -k "not cfg2" > gpurun_out/r05f/contract.log 2>&1; tail -5 gpurun_out/r05f/contract.log
for f in gpurun_out/r05f/u*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][0]
    r=j.get('roofline',{})
    print(j['value'], j['ms_per_step'], r.get('avg_launch_ms'), r.get('frac'), r.get('over_step'), j.get('stage_ms_one_batch_in_flight'), j['config']['accuracy_vs_synthetic_truth'], j.get('cpu_baseline',{}).get('verdict_agreement_with_gpu'))
except Exception as e: print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
