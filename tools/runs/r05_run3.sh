#!/bin/bash
# round 5, GPU run 3: frames per submitted unit (the step's 256 frames stay the workload; here the per-GPU batch itself is varied to find the shape)
cd $GRAFT_REPO_ROOT
REPS=2 bash tools/ab_matrix.sh usweep \
  'b144|||--steps 70 --warmup 8 --batch 144' \
  'b160|||--steps 64 --warmup 8 --batch 160' \
  'b176|||--steps 58 --warmup 8 --batch 176' \
  'b192|||--steps 54 --warmup 8 --batch 192' \
  'b208|||--steps 50 --warmup 6 --batch 208' \
  'b224|||--steps 46 --warmup 6 --batch 224' \
  'b240|||--steps 42 --warmup 6 --batch 240' \
  'b256|||--steps 40 --warmup 5' \
  'b288|||--steps 36 --warmup 5 --batch 288' \
  'b320|||--steps 32 --warmup 4 --batch 320' \
  'b512|||--steps 20 --warmup 3 --batch 512' > gpurun_out/r05c_usweep.txt 2>&1
cat gpurun_out/r05c_usweep.txt
