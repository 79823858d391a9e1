#!/bin/bash
# round 6, run 11: where do the 10 % of the collective path at world size 1 go?  launcher alone / + process group (nccl, gloo); kernel trace of the nccl form
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run11; mkdir -p $out
A="--steps 20 --warmup 5 --no-cpu-baseline --no-host-frames"
for rep in 1 2; do
python bench.py $A 2>/dev/null | tail -1 > $out/a_plain_$rep.json
SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py $A 2>/dev/null | tail -1 > $out/b_launcher_$rep.json
SLIDEO_BENCH_FORCE_LAUNCH=1 OMP_NUM_THREADS=16 python bench.py $A 2>/dev/null | tail -1 > $out/b2_launcher_omp16_$rep.json
SLIDEO_BENCH_FORCE_DIST=1 SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py $A 2>/dev/null | tail -1 > $out/c_nccl_$rep.json
SLIDEO_BENCH_FORCE_DIST=1 SLIDEO_BENCH_FORCE_LAUNCH=1 python bench.py $A --backend gloo 2>/dev/null | tail -1 > $out/d_gloo_$rep.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_run11/*.json')):
    try: j=json.load(open(f)); print(f.split('/')[-1], j['value'], j['ms_per_step'], j['config'].get('collective'))
    except Exception as e: print(f, 'FAILED', e)
PY
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SLIDEO_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 rocprofv3 --kernel-trace --stats -d $out/prof -o t -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-frames > $out/prof.log 2>&1
python profiles/summarize_rocpd.py $out/prof/t_results.db | head -40
rm -rf $out/prof
