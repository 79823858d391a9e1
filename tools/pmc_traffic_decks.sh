#!/bin/bash
# usage (on the GPU box): tools/pmc_traffic_decks.sh <tag>
# Fabric-side traffic of the search kernel at the deck sizes where the FP4 train matrix stops fitting the Infinity Cache comfortably
# (VERDICT r05 item 3): configs[3] (1000 pages x ORB-1000: ~100 MB of FP4 operand) and configs[4] (1000 pages x ORB-2000: ~210 MB), one
# unit in flight and four.  FETCH_SIZE / WRITE_SIZE in separate --pmc passes (KiB); FETCH is doubled by the reader (gfx950 correction).
tag=${1:-r06}
out=gpurun_out/traffic_$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() {  # name, bench args...
  n=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $out/pmc_${n}_$c -o t -- python bench.py "$@" --no-cpu-baseline --no-host-frames > $out/${n}_$c.log 2>&1
    python profiles/summarize_pmc.py $out/pmc_${n}_$c/t_results.db knn_tile > $out/${n}_$c.txt
    rm -rf $out/pmc_${n}_$c
  done
  tail -1 $out/${n}_FETCH_SIZE.log | python -c "
import sys,json
j=json.loads(sys.stdin.readline()); c=j['config']; r=j['roofline']
print('$n: Mu', c['train_descriptors_unique'], 'FP4 matrix MB', round(c['train_descriptors_unique']*128/1e6,1), 'pairs/launch', r['pairs_per_launch'], 'launches/step', r['launches_per_step'])"
  cat $out/${n}_FETCH_SIZE.txt $out/${n}_WRITE_SIZE.txt
}
run cfg3_alone   --workload cfg3 --total-frames 512 --steps 2 --warmup 1 --no-overlap
run cfg3_overlap --workload cfg3 --total-frames 2048 --steps 2 --warmup 1
run cfg4_alone   --workload cfg4 --steps 2 --warmup 1 --no-overlap
run cfg4_overlap --workload cfg4 --steps 4 --warmup 1
run headline_alone --steps 2 --warmup 1 --no-overlap
