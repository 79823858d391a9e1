#!/bin/bash
# round 6, after the unscaled matrix instruction: smoke, the driver-form bench, then everything profiles/ keeps (tools/final_profile.sh; the full GPU suite on this library: run 28)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final_r06
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_r06/smoke.log 2>&1; tail -1 gpurun_out/final_r06/smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/final_r06/bench_driver_form.json 2> gpurun_out/final_r06/bench_driver_form.err; tail -c 400 gpurun_out/final_r06/bench_driver_form.json
bash tools/final_profile.sh r06 > gpurun_out/final_r06/final_profile.log 2>&1
tail -3 gpurun_out/final_r06/final_profile.log
