#!/bin/bash
# round 6, run 21: the search's block shape while units share the chip, at the LARGE decks (configs[3]: 1000 pages; configs[4]: 1000 pages x ORB-2000), where the search is 3/4 of the work
cd $GRAFT_REPO_ROOT
REPS=2 tools/ab_env.sh r06_cfg3_share "--workload cfg3 --total-frames 20480 --steps 8 --warmup 2 --no-host-frames" base="" share0="SLIDEO_KNN_SHARE=0" w12="SLIDEO_KNN_SHARE=3" t1="SLIDEO_KNN_SHARE=5"
REPS=2 tools/ab_env.sh r06_cfg4_share "--workload cfg4 --steps 12 --warmup 4 --no-host-frames" base="" share0="SLIDEO_KNN_SHARE=0" w12="SLIDEO_KNN_SHARE=3" t1="SLIDEO_KNN_SHARE=5"
