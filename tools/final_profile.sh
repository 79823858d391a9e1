#!/bin/bash
# usage (on the GPU box): tools/final_profile.sh <tag>   -> everything profiles/ needs for one state of the code
tag=$1
out=gpurun_out/final_$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py > $out/bench_default.json 2> $out/bench_default.err
rocprofv3 --kernel-trace --stats -d $out/stats -o t -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap > $out/bench_no_overlap.json 2>&1
python profiles/summarize_rocpd.py $out/stats/t_results.db | grep -v rocclr > $out/kernel_stats_no_overlap.txt
rocprofv3 --kernel-trace --stats -d $out/stats2 -o t -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-host-frames > $out/bench_overlap.json 2>&1
python profiles/summarize_rocpd.py $out/stats2/t_results.db | grep -v rocclr > $out/kernel_stats_overlap.txt
python tools/timeline_overlap.py $out/stats2/t_results.db 100 > $out/timeline_overlap.txt
# the search with ONE block per CU (what overlapped units launch), alone on the chip
SLIDEO_KNN_SHARE=1 rocprofv3 --kernel-trace --stats -d $out/stats1b -o t -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap > $out/bench_no_overlap_one_block.json 2>&1
python profiles/summarize_rocpd.py $out/stats1b/t_results.db | grep -v rocclr | head -4 > $out/kernel_stats_no_overlap_one_block_per_cu.txt
rm -rf $out/stats1b
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap > $out/pmc_$c.log 2>&1
  python profiles/summarize_pmc.py $out/pmc_$c/t_results.db > $out/pmc_$c.txt
  # the same counter with 4 batches in flight (the timed configuration): what the kNN launch moves while ORB / verify kernels of
  # other batches share the L2 / Infinity Cache with it
  rocprofv3 --kernel-trace --pmc $c -d $out/pmcov_$c -o t -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-frames > $out/pmcov_$c.log 2>&1
  python profiles/summarize_pmc.py $out/pmcov_$c/t_results.db > $out/pmc_overlap_$c.txt
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $out/pmc_sq -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap > $out/pmc_sq.log 2>&1
python profiles/summarize_pmc.py $out/pmc_sq/t_results.db knn_tile > $out/pmc_sq_knn.txt
rm -rf $out/stats $out/stats2 $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmcov_FETCH_SIZE $out/pmcov_WRITE_SIZE $out/pmc_sq
{ python bench.py --workload cfg1 --steps 20 --warmup 4 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --workload cfg3 --steps 20 --warmup 2 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --workload cfg4 --steps 12 --warmup 4 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --workload tiny --steps 200 --warmup 20 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --workload cfg2 2>/dev/null | tail -1; } > $out/bench_other_workloads.jsonl
# the modes: the 8-DOF homography verifier (the sample solvers: the library's default since ABI 6 is hdlt 1; both verdict rules), the LSH-compatible index,
# configs[4] with either verifier, configs[2] with Lowe's test instead of its default tolerance vote
{ python bench.py --verify-model 1 --persp 0.1 --steps 20 --warmup 3 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --verify-model 1 --persp 0.1 --verdict-rule 1 --steps 20 --warmup 3 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --verify-model 1 --verdict-rule 1 --steps 20 --warmup 3 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --verify-model 0 --verdict-rule 1 --steps 20 --warmup 3 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --verify-model 1 --hdlt 0 --persp 0.1 --steps 6 --warmup 2 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --matcher lsh --steps 20 --warmup 3 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --workload cfg4 --verify-model 0 --persp 0 --steps 12 --warmup 4 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --workload cfg4 --hdlt 0 --steps 6 --warmup 2 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --workload cfg4 --hdlt 2 --steps 12 --warmup 4 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1
  python bench.py --workload cfg2 --sift-vote ratio --steps 20 --warmup 3 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1; } > $out/bench_modes.jsonl
rocprofv3 --kernel-trace --stats -d $out/stats3 -o t -- python bench.py --workload cfg2 --steps 3 --warmup 1 --no-cpu-baseline --no-host-frames > $out/bench_cfg2_under_rocprof.json 2>&1
python profiles/summarize_rocpd.py $out/stats3/t_results.db | grep -v rocclr > $out/kernel_stats_cfg2.txt
rocprofv3 --kernel-trace --stats -d $out/stats4 -o t -- python bench.py --verify-model 1 --persp 0.1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap > $out/bench_homography_under_rocprof.json 2>&1
python profiles/summarize_rocpd.py $out/stats4/t_results.db | grep -v rocclr > $out/kernel_stats_homography_hdlt1.txt
rocprofv3 --kernel-trace --stats -d $out/stats5 -o t -- python bench.py --verify-model 1 --hdlt 0 --persp 0.1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap > $out/bench_homography_hdlt0_under_rocprof.json 2>&1
python profiles/summarize_rocpd.py $out/stats5/t_results.db | grep -v rocclr > $out/kernel_stats_homography_hdlt0.txt
rocprofv3 --kernel-trace --stats -d $out/stats6 -o t -- python bench.py --matcher lsh --steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap > $out/bench_lsh_under_rocprof.json 2>&1
python profiles/summarize_rocpd.py $out/stats6/t_results.db | grep -v rocclr > $out/kernel_stats_lsh.txt
rm -rf $out/stats3 $out/stats4 $out/stats5 $out/stats6
tail -c 600 $out/bench_default.json; head -16 $out/kernel_stats_no_overlap.txt
python tools/group_rate.py > $out/group_rate.txt 2>/dev/null
python tools/hdlt_agreement.py > $out/hdlt_agreement.json 2>/dev/null
python tools/variant_sensitivity.py > $out/variant_sensitivity.json 2> $out/variant_sensitivity.err
python tools/host_path_rate.py > $out/host_path_rate.txt 2>/dev/null
bash tools/pmc_kernels.sh > /dev/null 2>&1; cp gpurun_out/pmc_kernels.txt $out/pmc_orb_verify_kernels.txt
python tools/stress_determinism.py 2000 > $out/stress_determinism.txt 2>&1
