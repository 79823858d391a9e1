#!/bin/bash
# round 6, run 16: kernel stats of configs[4] (one unit in flight) and configs[1]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run16; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/p4 -o t -- python bench.py --workload cfg4 --steps 4 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap > $out/cfg4.log 2>&1
python profiles/summarize_rocpd.py $out/p4/t_results.db | grep -v rocclr | head -32 > $out/cfg4_kernel_stats_no_overlap.txt; cat $out/cfg4_kernel_stats_no_overlap.txt
rocprofv3 --kernel-trace --stats -d $out/p1 -o t -- python bench.py --workload cfg1 --steps 4 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap > $out/cfg1.log 2>&1
python profiles/summarize_rocpd.py $out/p1/t_results.db | grep -v rocclr | head -24 > $out/cfg1_kernel_stats_no_overlap.txt; cat $out/cfg1_kernel_stats_no_overlap.txt
rm -rf $out/p4 $out/p1
