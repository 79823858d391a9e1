#!/bin/bash
# round 6, run 2: ring counters in inline assembly (no compiler vmcnt(0) in front of them) for both wave shapes; the 1-tile block with a lagged publish
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_run2
SLIDEO_KNN_SHARE=6 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn or blocks_per_cu or end_to_end or dedup or fused or ratio" > gpurun_out/r06_run2/parity_share6.log 2>&1; tail -2 gpurun_out/r06_run2/parity_share6.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn or blocks_per_cu or end_to_end or dedup or fused or ratio" > gpurun_out/r06_run2/parity_default.log 2>&1; tail -2 gpurun_out/r06_run2/parity_default.log
REPS=2 tools/ab_env.sh r06_t1b "--steps 100 --no-host-frames" base="" ringbuiltin="SLIDEO_LIB_PATH=slideo_amd/lib/variants/ringbuiltin/libslideo_amd.so" t1shared="SLIDEO_KNN_SHARE=5" t1always="SLIDEO_KNN_SHARE=6"
