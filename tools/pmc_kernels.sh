#!/bin/bash
# usage (on the GPU box): tools/pmc_kernels.sh  -> gpurun_out/pmc_kernels.txt: SQ counters of the ORB / verify kernels
# (two separate --pmc passes with --kernel-trace only), headline workload, one batch in flight
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_kernels; mkdir -p $out
A="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
B="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVES"
rocprofv3 --kernel-trace --pmc $A -d $out/a -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap > $out/a.log 2>&1
rocprofv3 --kernel-trace --pmc $B -d $out/b -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap > $out/b.log 2>&1
{
  echo "# rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-frames --no-overlap   (3 dispatches of the 256-frame kernels;"
  echo "# the ORB kernels also run for the page ingest).  VALU share of a kernel = SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * duration * clock)."
  echo "## set A: $A"
  for k in fast_kernel blur_f32_kernel describe_blurred_kernel resize_quad_kernel gray_kernel reproject ransac_kernel vote_kernel knn_expand_dups sort_kernel compact_kernel knn_tile; do python profiles/summarize_pmc.py $out/a/t_results.db $k; done
  echo "## set B: $B"
  for k in fast_kernel blur_f32_kernel describe_blurred_kernel resize_quad_kernel gray_kernel reproject ransac_kernel vote_kernel knn_expand_dups sort_kernel compact_kernel knn_tile; do python profiles/summarize_pmc.py $out/b/t_results.db $k; done
} > gpurun_out/pmc_kernels.txt
rm -rf $out
tail -5 gpurun_out/pmc_kernels.txt
