#!/bin/bash
# round 6, run 7: the tile half norms fetched one super-tile ahead (scalar load off the first chain's path): A/B, and the 1-tile block's probe
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run7; mkdir -p $out
SLIDEO_LIB_PATH=slideo_amd/lib/variants/nmahead/libslideo_amd.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn or blocks_per_cu or end_to_end or dedup or fused or ratio" > $out/parity.log 2>&1; tail -2 $out/parity.log
REPS=3 tools/ab_env.sh r06_nmahead "--steps 100 --no-host-frames" base="" nmahead="SLIDEO_LIB_PATH=slideo_amd/lib/variants/nmahead/libslideo_amd.so"
n=nmaheadprobe
SLIDEO_KNN_SHARE=6 SLIDEO_LIB_PATH=slideo_amd/lib/variants/$n/libslideo_amd.so python bench.py --steps 6 --warmup 2 --no-overlap --no-cpu-baseline --no-host-frames 2>$out/${n}_alone.err | tail -1 > $out/${n}_alone.json
grep KT_PROBE $out/${n}_alone.err | tail -2
SLIDEO_KNN_SHARE=1 SLIDEO_LIB_PATH=slideo_amd/lib/variants/$n/libslideo_amd.so python bench.py --steps 6 --warmup 2 --no-overlap --no-cpu-baseline --no-host-frames 2>$out/${n}_t2alone.err | tail -1 > $out/${n}_t2alone.json
grep KT_PROBE $out/${n}_t2alone.err | tail -2
