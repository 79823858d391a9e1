#!/bin/bash
# round 6, run 29: the K-split chains (-DKT_KSPLIT: k0 -> k2 and k1 -> k3 in two accumulators per query tile, summed in front of the tree; 146 registers,
# two waves per SIMD): parity on the variant, A/B headline / configs[1] / one block per CU alone
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run29; mkdir -p $out
V=slideo_amd/lib/variants/ksplit/libslideo_amd.so
SLIDEO_LIB_PATH=$V timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn or end_to_end or dedup or fused or ratio or modes" > $out/parity.log 2>&1; tail -2 $out/parity.log
REPS=3 tools/ab_env.sh r06_ksplit "--steps 60 --no-host-frames" base="" ksplit="SLIDEO_LIB_PATH=$V"
REPS=2 tools/ab_env.sh r06_ksplit_1b "--steps 8 --warmup 2 --no-host-frames --no-overlap" base="SLIDEO_KNN_SHARE=1" ksplit="SLIDEO_KNN_SHARE=1 SLIDEO_LIB_PATH=$V"
REPS=1 tools/ab_env.sh r06_ksplit_cfg1 "--workload cfg1 --steps 60 --no-host-frames" base="" ksplit="SLIDEO_LIB_PATH=$V"
