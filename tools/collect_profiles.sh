#!/bin/bash
# usage (here, after tools/final_profile.sh <tag> ran on the GPU box): tools/collect_profiles.sh <tag> <round prefix, e.g. r05>
# copies gpurun_out/final_<tag>/ into profiles/ under the names profiles/README.md uses
s=gpurun_out/final_$1; r=$2
cp $s/bench_default.json profiles/${r}_bench_headline.json
cp $s/kernel_stats_no_overlap.txt profiles/${r}_bench_headline_kernel_stats_no_overlap.txt
cp $s/kernel_stats_no_overlap_one_block_per_cu.txt profiles/${r}_bench_headline_kernel_stats_no_overlap_one_block_per_cu.txt
cp $s/kernel_stats_overlap.txt profiles/${r}_bench_headline_kernel_stats_overlap.txt
tail -1 $s/bench_no_overlap.json > profiles/${r}_bench_headline_under_rocprof_no_overlap.json
tail -1 $s/bench_overlap.json > profiles/${r}_bench_headline_under_rocprof_overlap.json
cp $s/bench_modes.jsonl profiles/${r}_bench_modes.jsonl
cp $s/bench_other_workloads.jsonl profiles/${r}_bench_other_workloads.jsonl
cp $s/kernel_stats_cfg2.txt profiles/${r}_cfg2_sift_l2_kernel_stats.txt
cp $s/kernel_stats_homography_hdlt0.txt profiles/${r}_homography_hdlt0_kernel_stats.txt
cp $s/kernel_stats_homography_hdlt1.txt profiles/${r}_homography_hdlt1_kernel_stats.txt
cp $s/kernel_stats_lsh.txt profiles/${r}_lsh_kernel_stats.txt
cp $s/group_rate.txt profiles/${r}_group_rate.txt
cp $s/host_path_rate.txt profiles/${r}_host_path_rate.txt
cp $s/hdlt_agreement.json profiles/${r}_hdlt_agreement.json
cp $s/variant_sensitivity.json profiles/${r}_variant_sensitivity.json
cp $s/stress_determinism.txt profiles/${r}_stress_determinism.txt
cp $s/timeline_overlap.txt profiles/${r}_timeline_overlap.txt
cp $s/pmc_orb_verify_kernels.txt profiles/${r}_pmc_orb_verify_kernels.txt
cp $s/pmc_sq_knn.txt profiles/${r}_pmc_knn_tile2.txt
{ echo "# FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes; KiB, FETCH x2-corrected where the summary says so), one batch in flight, then four";
  for f in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_overlap_FETCH_SIZE pmc_overlap_WRITE_SIZE; do echo "## $f"; cat $s/$f.txt; done; } > profiles/${r}_pmc_hbm_traffic.txt
ls profiles | grep "^${r}_"
