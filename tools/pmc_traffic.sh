# usage (on the GPU box): tools/pmc_traffic.sh  -> the four FETCH_SIZE / WRITE_SIZE passes of tools/final_profile.sh alone (after a change of knn_tile.hip.h: profiles/knn_traffic.json is tied to its sha1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/final_${1:-r06}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap > $out/pmc_$c.log 2>&1
  python profiles/summarize_pmc.py $out/pmc_$c/t_results.db > $out/pmc_$c.txt
  rocprofv3 --kernel-trace --pmc $c -d $out/pmcov_$c -o t -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-frames > $out/pmcov_$c.log 2>&1
  python profiles/summarize_pmc.py $out/pmcov_$c/t_results.db > $out/pmc_overlap_$c.txt
  rm -rf $out/pmc_$c $out/pmcov_$c
done
python bench.py > $out/bench_default.json 2> $out/bench_default.err
grep -A2 "knn_tile2_kernel" $out/pmc_FETCH_SIZE.txt | head -3
