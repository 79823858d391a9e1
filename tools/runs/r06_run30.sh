#!/bin/bash
# round 6, run 30: block shapes of the shared mode re-checked on the unscaled instruction (headline; 800 pages = either side of the 290-pairs-per-pixel rule)
cd $GRAFT_REPO_ROOT
REPS=2 tools/ab_env.sh r06_u_shapes "--steps 60 --no-host-frames" rule="" w12="SLIDEO_KNN_SHARE=3" two="SLIDEO_KNN_SHARE=0"
REPS=2 tools/ab_env.sh r06_u_700 "--pages 700 --steps 40 --no-host-frames" w8="SLIDEO_KNN_W12_RATIO=0" w12="SLIDEO_KNN_SHARE=3"
