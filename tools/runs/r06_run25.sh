#!/bin/bash
# round 6, run 25: configs[3] (12-wave search block: 128 registers per SIMD lane left): fast_kernel / blur_f32_kernel capped at <= 64 registers (spilling) — two of their waves per SIMD instead of one
cd $GRAFT_REPO_ROOT
REPS=2 tools/ab_env.sh r06_cfg3_caps "--workload cfg3 --total-frames 20480 --steps 8 --warmup 2 --no-host-frames" base="" fast60="SLIDEO_LIB_PATH=slideo_amd/lib/variants/fast60/libslideo_amd.so" fastblur60="SLIDEO_LIB_PATH=slideo_amd/lib/variants/fastblur60/libslideo_amd.so"
REPS=1 tools/ab_env.sh r06_head_caps "--steps 40 --warmup 4 --no-host-frames" base="" fast60="SLIDEO_LIB_PATH=slideo_amd/lib/variants/fast60/libslideo_amd.so" fast60w12="SLIDEO_KNN_SHARE=3 SLIDEO_LIB_PATH=slideo_amd/lib/variants/fast60/libslideo_amd.so" fastblur60w12="SLIDEO_KNN_SHARE=3 SLIDEO_LIB_PATH=slideo_amd/lib/variants/fastblur60/libslideo_amd.so"
