#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05g
timeout 1500 python -m pytest tests/test_gpu_sift.py tests/test_gpu_sift_matcher.py -m gpu -q -x > gpurun_out/r05g/sift_tests.log 2>&1; tail -3 gpurun_out/r05g/sift_tests.log
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k sift > gpurun_out/r05g/sift_fuzz.log 2>&1; tail -2 gpurun_out/r05g/sift_fuzz.log
for P in 1 2 4 8; do
  SLIDEO_SIFT_PASSES=$P timeout 600 python bench.py --workload cfg2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r05g/cfg2_p$P.json 2> gpurun_out/r05g/cfg2_p$P.err
done
timeout 600 python bench.py --workload cfg2 --steps 8 --warmup 2 > gpurun_out/r05g/cfg2_default.json 2> gpurun_out/r05g/cfg2_default.err
for f in gpurun_out/r05g/cfg2_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][0]
    print(j['value'], j['ms_per_step'], j['stage_ms_per_batch'], j['config']['accuracy_vs_synthetic_truth'], j['config']['checked'])
except Exception as e: print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
