#!/bin/bash
# round 6, run 22: where the 12-wave block of the 2-tile wave starts to win over the 8-wave block while units share the chip: deck sizes between the headline's and configs[3]'s
cd $GRAFT_REPO_ROOT
for P in 600 700 800; do
REPS=2 tools/ab_env.sh r06_w12_p$P "--pages $P --steps 40 --warmup 4 --no-host-frames" base="" w12="SLIDEO_KNN_SHARE=3"
done
