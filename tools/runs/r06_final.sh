#!/bin/bash
# round 6: the whole GPU suite, the driver-form bench (--steps 20 --warmup 5), then everything profiles/ keeps (tools/final_profile.sh)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final_r06
timeout 3300 python -m pytest tests -m gpu -q > gpurun_out/final_r06/tests_full.log 2>&1; tail -4 gpurun_out/final_r06/tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_r06/smoke.log 2>&1; tail -1 gpurun_out/final_r06/smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/final_r06/bench_driver_form.json 2> gpurun_out/final_r06/bench_driver_form.err; tail -c 400 gpurun_out/final_r06/bench_driver_form.json
bash tools/final_profile.sh r06 > gpurun_out/final_r06/final_profile.log 2>&1
tail -3 gpurun_out/final_r06/final_profile.log
