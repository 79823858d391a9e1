#!/bin/bash
# round 6, run 9: configs[1] (ORB + verify bound): the pipeline's switches
cd $GRAFT_REPO_ROOT
REPS=2 tools/ab_env.sh r06_cfg1_sw "--workload cfg1 --steps 60 --no-host-frames" base="" nochain="SLIDEO_ORB_CHAIN=0" share0="SLIDEO_KNN_SHARE=0" share0nochain="SLIDEO_KNN_SHARE=0 SLIDEO_ORB_CHAIN=0" unit128="SLIDEO_X=1"
REPS=1 tools/ab_env.sh r06_cfg1_unit "--workload cfg1 --steps 60 --no-host-frames --unit 128" u128="" u128nochain="SLIDEO_ORB_CHAIN=0"
REPS=1 tools/ab_env.sh r06_cfg1_unit64 "--workload cfg1 --steps 60 --no-host-frames --unit 64" u64="" u64nochain="SLIDEO_ORB_CHAIN=0"
