#!/bin/bash
# round 6, run 36: the bound-only prologue of the exact search (KT_PRO_V super-tiles visited twice: first without lists, to tighten the thresholds): parity, A/B by length
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run36; mkdir -p $out
V=slideo_amd/lib/variants
SLIDEO_LIB_PATH=$V/pro32/libslideo_amd.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $out/parity.log 2>&1; tail -2 $out/parity.log
REPS=3 tools/ab_env.sh r06_pro "--steps 60 --no-host-frames" pro0="SLIDEO_LIB_PATH=$V/pro0/libslideo_amd.so" pro16="SLIDEO_LIB_PATH=$V/pro16/libslideo_amd.so" pro32="SLIDEO_LIB_PATH=$V/pro32/libslideo_amd.so" pro64="SLIDEO_LIB_PATH=$V/pro64/libslideo_amd.so"
REPS=2 tools/ab_env.sh r06_pro_1b "--steps 8 --warmup 2 --no-host-frames --no-overlap" pro0="SLIDEO_KNN_SHARE=1 SLIDEO_LIB_PATH=$V/pro0/libslideo_amd.so" pro32="SLIDEO_KNN_SHARE=1 SLIDEO_LIB_PATH=$V/pro32/libslideo_amd.so"
REPS=1 tools/ab_env.sh r06_pro_cfg3 "--workload cfg3 --total-frames 20480 --steps 8 --warmup 2 --no-host-frames" pro0="SLIDEO_LIB_PATH=$V/pro0/libslideo_amd.so" pro32="SLIDEO_LIB_PATH=$V/pro32/libslideo_amd.so"
REPS=1 tools/ab_env.sh r06_pro_cfg1 "--workload cfg1 --steps 60 --no-host-frames" pro0="SLIDEO_LIB_PATH=$V/pro0/libslideo_amd.so" pro32="SLIDEO_LIB_PATH=$V/pro32/libslideo_amd.so"
