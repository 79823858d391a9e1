#!/bin/bash
# round 5, GPU run 4: where fast_kernel's time goes (ablation builds, timing only) + rocprof of the product build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
for a in base 0 1 2 3; do
  if [ $a = base ]; then fl=""; else fl="-DFAST_ABL=$a"; fi
  SLIDEO_HIP_EXTRA_FLAGS="$fl" python -m slideo_amd.build --tag abl_$a > gpurun_out/r05d/build_$a.log 2>&1
  mkdir -p gpurun_out/r05d/prof_$a; (cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
  SLIDEO_LIB_PATH=slideo_amd/lib/variants/abl_$a/libslideo_amd.so rocprofv3 --kernel-trace --stats -d gpurun_out/r05d/prof_$a -o t -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-overlap > gpurun_out/r05d/prof_$a/bench.log 2>&1)
  echo "== FAST_ABL $a"; python profiles/summarize_rocpd.py gpurun_out/r05d/prof_$a/t_results.db | grep -E "fast_kernel|resize_kernel|gray_kernel|describe|blur_f32|kernel  " | head -8
  rm -f gpurun_out/r05d/prof_$a/t_results.db
done
