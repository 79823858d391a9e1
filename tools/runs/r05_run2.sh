#!/bin/bash
# round 5, GPU run 2: instruction priority of the ORB / re-projection waves beside the search; smaller units; the search's per-wave time breakdown
cd $GRAFT_REPO_ROOT
REPS=2 bash tools/ab_matrix.sh prio \
  'base|||' \
  'orb1|-DORB_PRIO=1||' \
  'orb2|-DORB_PRIO=2||' \
  'orb3|-DORB_PRIO=3||' \
  'vt1|-DVERIFY_PRIO=1||' \
  'vt2|-DVERIFY_PRIO=2||' \
  'orb2vt1|-DORB_PRIO=2 -DVERIFY_PRIO=1||' \
  'orb3vt3|-DORB_PRIO=3 -DVERIFY_PRIO=3||' > gpurun_out/r05b_prio.txt 2>&1
REPS=1 bash tools/ab_matrix.sh units \
  'b256|||--steps 40 --warmup 5' \
  'b128|||--steps 80 --warmup 10 --batch 128' \
  'b64|||--steps 160 --warmup 20 --batch 64' \
  'b192|||--steps 60 --warmup 8 --batch 192' \
  'b384|||--steps 30 --warmup 4 --batch 384' > gpurun_out/r05b_units.txt 2>&1
REPS=1 bash tools/ab_matrix.sh probe \
  'probe_share1_alone|-DKT_PROBE|SLIDEO_KNN_SHARE=1|--steps 10 --warmup 2 --no-overlap' \
  'probe_share0_alone|-DKT_PROBE|SLIDEO_KNN_SHARE=0|--steps 10 --warmup 2 --no-overlap' \
  'probe_overlap|-DKT_PROBE||--steps 20 --warmup 3' > gpurun_out/r05b_probe.txt 2>&1
grep -h KT_ gpurun_out/abm_probe/*.err >> gpurun_out/r05b_probe.txt
cat gpurun_out/r05b_prio.txt gpurun_out/r05b_units.txt
tail -30 gpurun_out/r05b_probe.txt
