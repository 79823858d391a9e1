#!/bin/bash
# round 6, run 6: the 1-tile block with every fragment re-load right behind the MFMA that read its registers (-DKT1_FENCED), probe build
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run6; mkdir -p $out
for n in t1fenced t1probe; do
  echo "== $n alone (one unit in flight)"
  SLIDEO_KNN_SHARE=6 SLIDEO_LIB_PATH=slideo_amd/lib/variants/$n/libslideo_amd.so python bench.py --steps 6 --warmup 2 --no-overlap --no-cpu-baseline --no-host-frames 2>$out/${n}_alone.err | tail -1 > $out/${n}_alone.json
  grep KT_PROBE $out/${n}_alone.err | tail -2
  python -c "import json;j=json.load(open('$out/${n}_alone.json'));print('ms/step',j['ms_per_step'],j['stage_ms_per_step'], 'acc', j['config']['accuracy_vs_synthetic_truth'])"
  echo "== $n shared (four units in flight)"
  SLIDEO_KNN_SHARE=6 SLIDEO_LIB_PATH=slideo_amd/lib/variants/$n/libslideo_amd.so python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-host-frames 2>$out/${n}_shared.err | tail -1 > $out/${n}_shared.json
  grep KT_PROBE $out/${n}_shared.err | tail -2
  python -c "import json;j=json.load(open('$out/${n}_shared.json'));print('ms/step',j['ms_per_step'],j['stage_ms_per_step'],j['roofline']['per_launch']['avg_launch_ms'], 'acc', j['config']['accuracy_vs_synthetic_truth'])"
done
