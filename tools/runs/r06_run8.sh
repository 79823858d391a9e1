#!/bin/bash
# round 6, run 8: fused multiply-adds in reproject_vt_kernel's area sums: parity (tolerance 1e-4), the largest |d similarity| at headline size, A/B
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run8; mkdir -p $out
V=slideo_amd/lib/variants/vtfma/libslideo_amd.so
SLIDEO_LIB_PATH=$V timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_homography.py -x -q -m gpu > $out/parity.log 2>&1; tail -2 $out/parity.log
SLIDEO_LIB_PATH=$V timeout 900 python -m pytest tests/test_gpu_big_shapes.py -x -q -m gpu -k "headline_shape or configs1_exact or configs4" > $out/big.log 2>&1; tail -2 $out/big.log
REPS=3 tools/ab_env.sh r06_vtfma "--steps 100 --no-host-frames" base="" vtfma="SLIDEO_LIB_PATH=$V"
REPS=2 tools/ab_env.sh r06_vtfma_cfg1 "--workload cfg1 --steps 60 --no-host-frames" base="" vtfma="SLIDEO_LIB_PATH=$V"
REPS=1 tools/ab_env.sh r06_vtfma_cfg4 "--workload cfg4 --steps 12 --warmup 4 --no-host-frames" base="" vtfma="SLIDEO_LIB_PATH=$V"
