#!/bin/bash
# round 6, run 24: configs[3] with the 12-wave block: frames per unit (318 blocks of 768 queries = 1.24 rounds at 256 frames; 192 frames = 239 blocks: one round)
cd $GRAFT_REPO_ROOT
for U in 256 192 204 128; do
REPS=2 tools/ab_env.sh r06_cfg3_u$U "--workload cfg3 --total-frames 20480 --steps 8 --warmup 2 --no-host-frames --unit $U" u$U=""
done
