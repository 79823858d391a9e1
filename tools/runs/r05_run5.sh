#!/bin/bash
# round 5, GPU run 5: the 12-wave search block (three waves per SIMD): parity under SLIDEO_KNN_SHARE=3, then the bench A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
SLIDEO_KNN_SHARE=3 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "knn or end_to_end or dedup or fused" > gpurun_out/r05e/parity_w12.log 2>&1; tail -3 gpurun_out/r05e/parity_w12.log
SLIDEO_KNN_SHARE=3 timeout 1500 python -m pytest tests/test_gpu_big_shapes.py -m gpu -q -x -k "headline_shape_traces" > gpurun_out/r05e/big_w12.log 2>&1; tail -3 gpurun_out/r05e/big_w12.log
REPS=2 bash tools/ab_matrix.sh w12 \
  'base|||' \
  'w12shared||SLIDEO_KNN_SHARE=2|' \
  'w12always||SLIDEO_KNN_SHARE=3|' \
  'base192|||--steps 54 --warmup 8 --batch 192' \
  'w12shared192||SLIDEO_KNN_SHARE=2|--steps 54 --warmup 8 --batch 192' \
  'w12shared208||SLIDEO_KNN_SHARE=2|--steps 50 --warmup 8 --batch 208' \
  'w12shared224||SLIDEO_KNN_SHARE=2|--steps 46 --warmup 8 --batch 224' \
  'w12shared160||SLIDEO_KNN_SHARE=2|--steps 64 --warmup 8 --batch 160' \
  'w12alone||SLIDEO_KNN_SHARE=3|--steps 10 --warmup 2 --no-overlap' > gpurun_out/r05e/ab.txt 2>&1
cat gpurun_out/r05e/ab.txt
