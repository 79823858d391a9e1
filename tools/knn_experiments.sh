#!/bin/bash
# Where does the kNN kernel's time go?  Experiment builds of the same sources (never shipped: results are wrong by design in
# the first one), each timed with bench.py --no-overlap (kernel alone on the GPU).  Build them in the CPU container:
#   tools/knn_experiments.sh build      -> slideo_amd/lib/exp_*.so
# and run on the GPU box:
#   tools/knn_experiments.sh run        -> gpurun_out/knn_experiments.txt
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-gpu-rdc -I include -I slideo_amd/csrc"
declare -A V=( [ring3]="-DKT_RING_V=3 -DKT_AHEAD_V=1" [ring3_v104]="-DKT_RING_V=3 -DKT_AHEAD_V=1 -DKT2_VGPRS=52" [ring3_v96]="-DKT_RING_V=3 -DKT_AHEAD_V=1 -DKT2_VGPRS=48" [v104]="-DKT2_VGPRS=52" )
if [ "$1" = build ]; then
  for k in "${!V[@]}"; do hipcc $FLAGS ${V[$k]} -o slideo_amd/lib/exp_$k.so slideo_amd/csrc/slideo_capi.hip & done; wait
  ls -la slideo_amd/lib/exp_*.so
else
  out=gpurun_out/knn_experiments.txt; : > $out
  for k in base "${!V[@]}" base; do
    if [ $k = base ]; then unset SLIDEO_LIB_PATH; else export SLIDEO_LIB_PATH=$PWD/slideo_amd/lib/exp_$k.so; fi
    timeout 90 python bench.py --steps 20 --warmup 4 --no-cpu-baseline ${KNN_EXP_BENCH_ARGS:---no-overlap} 2>/dev/null | python -c "
import json,sys
l=sys.stdin.readline()
if not l.strip(): print('$k', 'FAILED or timed out'); sys.exit(0)
d=json.loads(l); print('$k', 'knn_ms', d['roofline']['avg_launch_ms'], 'alone', (d.get('stage_ms_one_batch_in_flight') or {}).get('knn'), 'step_ms', d['ms_per_step'], 'acc', d['config']['accuracy_vs_synthetic_truth'])" >> $out
  done
  cat $out
fi
