#!/bin/bash
# usage (on the GPU box): tools/final_profile.sh <tag>   -> everything profiles/ needs for one state of the code
tag=$1
out=gpurun_out/final_$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py > $out/bench_default.json 2> $out/bench_default.err
rocprofv3 --kernel-trace --stats -d $out/stats -o t -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-overlap > $out/bench_no_overlap.json 2>&1
python profiles/summarize_rocpd.py $out/stats/t_results.db | grep -v rocclr > $out/kernel_stats_no_overlap.txt
rocprofv3 --kernel-trace --stats -d $out/stats2 -o t -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/bench_overlap.json 2>&1
python profiles/summarize_rocpd.py $out/stats2/t_results.db | grep -v rocclr > $out/kernel_stats_overlap.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap > $out/pmc_$c.log 2>&1
  python profiles/summarize_pmc.py $out/pmc_$c/t_results.db > $out/pmc_$c.txt
  # the same counter with 4 batches in flight (the timed configuration): what the kNN launch moves while ORB / verify kernels of
  # other batches share the L2 / Infinity Cache with it
  rocprofv3 --kernel-trace --pmc $c -d $out/pmcov_$c -o t -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/pmcov_$c.log 2>&1
  python profiles/summarize_pmc.py $out/pmcov_$c/t_results.db > $out/pmc_overlap_$c.txt
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $out/pmc_sq -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap > $out/pmc_sq.log 2>&1
python profiles/summarize_pmc.py $out/pmc_sq/t_results.db knn_tile > $out/pmc_sq_knn.txt
rm -rf $out/stats $out/stats2 $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmcov_FETCH_SIZE $out/pmcov_WRITE_SIZE $out/pmc_sq
{ python bench.py --workload cfg1 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1
  python bench.py --workload cfg4 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1
  python bench.py --workload tiny --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1
  python bench.py --workload cfg2 2>/dev/null | tail -1; } > $out/bench_other_workloads.jsonl
tail -c 600 $out/bench_default.json; head -16 $out/kernel_stats_no_overlap.txt
