#!/bin/bash
# round 6, run 5: the split schedule of the 2-tile search wave (-DKT_SPLIT_V=1): parity, then the step A/B
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run5; mkdir -p $out
SLIDEO_LIB_PATH=slideo_amd/lib/variants/split/libslideo_amd.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn or blocks_per_cu or end_to_end or dedup or fused or ratio" > $out/parity_split.log 2>&1; tail -3 $out/parity_split.log
SLIDEO_LIB_PATH=slideo_amd/lib/variants/split/libslideo_amd.so timeout 600 python -m pytest tests/test_gpu_big_shapes.py -x -q -m gpu -k "headline_shape_traces" > $out/headline_split.log 2>&1; tail -3 $out/headline_split.log
REPS=3 tools/ab_env.sh r06_split "--steps 100 --no-host-frames" base="" split="SLIDEO_LIB_PATH=slideo_amd/lib/variants/split/libslideo_amd.so"
REPS=2 tools/ab_env.sh r06_split_cfg1 "--workload cfg1 --steps 60 --no-host-frames" base="" split="SLIDEO_LIB_PATH=slideo_amd/lib/variants/split/libslideo_amd.so"
