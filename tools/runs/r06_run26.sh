#!/bin/bash
# round 6, run 26: train-stream segments of the search at configs[4] (12-wave block: 167 query blocks for 256 CUs) and configs[3] / headline for reference
cd $GRAFT_REPO_ROOT
REPS=2 tools/ab_env.sh r06_cfg4_nseg "--workload cfg4 --steps 12 --warmup 4 --no-host-frames" plan="" n1="SLIDEO_KNN_NSEG=1" n2="SLIDEO_KNN_NSEG=2" n3="SLIDEO_KNN_NSEG=3"
REPS=1 tools/ab_env.sh r06_cfg3_nseg "--workload cfg3 --total-frames 20480 --steps 8 --warmup 2 --no-host-frames" plan="" n2="SLIDEO_KNN_NSEG=2" n4="SLIDEO_KNN_NSEG=4"
REPS=1 tools/ab_env.sh r06_head_nseg "--steps 40 --warmup 4 --no-host-frames" plan="" n2="SLIDEO_KNN_NSEG=2"
for f in gpurun_out/abe_r06_cfg4_nseg/plan_1.err; do tail -2 $f; done
