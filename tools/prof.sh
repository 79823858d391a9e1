#!/bin/bash
# usage (on the GPU box): tools/prof.sh <tag> [bench args]  -> per-kernel summary of one bench run
tag=$1; shift
mkdir -p gpurun_out/prof_$tag; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-frames "$@" > gpurun_out/prof_$tag/bench.log 2>&1
python profiles/summarize_rocpd.py gpurun_out/prof_$tag/t_results.db | grep -v rocclr | head -${PROF_LINES:-14}
grep -o '"value": [0-9.]*' gpurun_out/prof_$tag/bench.log | head -1
