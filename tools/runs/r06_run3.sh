#!/bin/bash
# round 6, run 3: where the 1-tile block's time goes (-DKT_PROBE), with and without its LDS-DMA fetches, publish lag 1 / 2 / 3
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run3; mkdir -p $out
for n in t1probe t1nodma t1lag1 t1lag3; do
  echo "== $n alone (one unit in flight)"
  SLIDEO_KNN_SHARE=6 SLIDEO_LIB_PATH=slideo_amd/lib/variants/$n/libslideo_amd.so python bench.py --steps 6 --warmup 2 --no-overlap --no-cpu-baseline --no-host-frames 2>$out/${n}_alone.err | tail -1 > $out/${n}_alone.json
  grep KT_PROBE $out/${n}_alone.err | tail -2
  python -c "import json;j=json.load(open('$out/${n}_alone.json'));print('ms/step',j['ms_per_step'],j['stage_ms_per_step'], 'acc', j['config']['accuracy_vs_synthetic_truth'])"
  echo "== $n shared (four units in flight)"
  SLIDEO_KNN_SHARE=6 SLIDEO_LIB_PATH=slideo_amd/lib/variants/$n/libslideo_amd.so python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-host-frames 2>$out/${n}_shared.err | tail -1 > $out/${n}_shared.json
  grep KT_PROBE $out/${n}_shared.err | tail -2
  python -c "import json;j=json.load(open('$out/${n}_shared.json'));print('ms/step',j['ms_per_step'],j['stage_ms_per_step'],j['roofline']['per_launch']['avg_launch_ms'], 'acc', j['config']['accuracy_vs_synthetic_truth'])"
done
