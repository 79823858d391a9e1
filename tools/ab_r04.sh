cd $GRAFT_REPO_ROOT; out=gpurun_out/prof1; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --workload cfg4 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > $out/cfg4_hdlt1.json
python bench.py --workload cfg4 --hdlt 2 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > $out/cfg4_hdlt2.json
python bench.py --workload cfg2 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $out/cfg2_tol.json
rocprofv3 --kernel-trace --stats -d $out/s1 -o t -- python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > $out/cfg4_rocprof.log 2>&1
python profiles/summarize_rocpd.py $out/s1/t_results.db | grep -v rocclr > $out/kernel_stats_cfg4_hdlt1_no_overlap.txt
rocprofv3 --kernel-trace --stats -d $out/s2 -o t -- python bench.py --workload cfg2 --steps 3 --warmup 1 --no-cpu-baseline > $out/cfg2_rocprof.log 2>&1
python profiles/summarize_rocpd.py $out/s2/t_results.db | grep -v rocclr > $out/kernel_stats_cfg2_tol.txt
rm -rf $out/s1 $out/s2
python - <<'PY'
import json
for f in ('cfg4_hdlt1','cfg4_hdlt2','cfg2_tol'):
    j=json.load(open('gpurun_out/prof1/%s.json'%f)); print(f, j['value'], j['ms_per_step'], j.get('stage_ms_per_step') or j.get('stage_ms_per_batch'), j.get('stage_ms_one_batch_in_flight'), j['config'].get('accuracy_vs_synthetic_truth'))
PY
head -32 $out/kernel_stats_cfg4_hdlt1_no_overlap.txt; head -30 $out/kernel_stats_cfg2_tol.txt
