#!/bin/bash
# round 6, run 33: the other group's tree behind the chain's SECOND MFMA (-DKT_LATE_TREE=4 / 6: 2 MFMAs, n VALU, 1 MFMA, n VALU, 1 MFMA) against the
# shipped (1 MFMA, 2 VALU) x 4; tools/mfma_order_probe.hip: 18.8 against 21.4 ns per MFMA at one wave per SIMD, equal at two
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run33; mkdir -p $out
V4=slideo_amd/lib/variants/late4/libslideo_amd.so; V6=slideo_amd/lib/variants/late6/libslideo_amd.so
SLIDEO_LIB_PATH=$V4 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn or end_to_end or dedup or fused" > $out/parity.log 2>&1; tail -1 $out/parity.log
REPS=3 tools/ab_env.sh r06_late "--steps 60 --no-host-frames" base="" late4="SLIDEO_LIB_PATH=$V4" late6="SLIDEO_LIB_PATH=$V6"
REPS=2 tools/ab_env.sh r06_late_1b "--steps 8 --warmup 2 --no-host-frames --no-overlap" base="SLIDEO_KNN_SHARE=1" late4="SLIDEO_KNN_SHARE=1 SLIDEO_LIB_PATH=$V4" late6="SLIDEO_KNN_SHARE=1 SLIDEO_LIB_PATH=$V6"
