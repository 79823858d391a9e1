cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab1
for rep in 1 2; do for p in 0 1 2; do
  SLIDEO_KNN_PRIO=$p python bench.py --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab1/prio${p}_$rep.json
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab1/*.json')):
    j=json.load(open(f)); r=j['roofline']
    print(f.split('/')[-1], j['value'], j['ms_per_step'], 'knn launch', r['avg_launch_ms'], 'frac', r['frac'], 'alone', r['one_batch_in_flight']['avg_launch_ms'], j['stage_ms_per_step'])
PY
python tools/hdlt_agreement.py > gpurun_out/r04_hdlt_agreement.json 2> gpurun_out/r04_hdlt_agreement.err; cat gpurun_out/r04_hdlt_agreement.json
python -m pytest tests/test_gpu_homography.py -q -x -k "sample_solver_forms" 2>&1 | tail -3
