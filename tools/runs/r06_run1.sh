#!/bin/bash
# round 6, run 1: the 1-tile 12-wave search block (knn_tile1.hip.h): parity under SLIDEO_KNN_SHARE=6, then the headline step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_run1
SLIDEO_KNN_SHARE=6 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn or blocks_per_cu or end_to_end or dedup or fused or ratio" > gpurun_out/r06_run1/parity_share6.log 2>&1; tail -3 gpurun_out/r06_run1/parity_share6.log
SLIDEO_KNN_SHARE=6 timeout 600 python -m pytest tests/test_gpu_big_shapes.py -x -q -m gpu -k "headline_shape_traces" > gpurun_out/r06_run1/headline_share6.log 2>&1; tail -3 gpurun_out/r06_run1/headline_share6.log
REPS=2 tools/ab_env.sh r06_t1 "--steps 100" base="" t1shared="SLIDEO_KNN_SHARE=5" t1always="SLIDEO_KNN_SHARE=6" w12="SLIDEO_KNN_SHARE=3"
