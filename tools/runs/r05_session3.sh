#!/bin/bash
# The gpurun calls of the round's third session (the instruction diet of the co-runners; profiles/r05_experiments.txt 7 and 8).
# Each line is the command of ONE `gpurun -- '<command>'` call, in order; variants are libraries built beforehand with
# `SLIDEO_HIP_EXTRA_FLAGS=... python -m slideo_amd.build --tag <name>` (slideo_amd/lib/variants/<name>/), "start" = the round's
# first commit (6c192ae) built in a git worktree, "prev" = the product library before the change under test.
# s1   python -m pytest tests -m gpu -x -q; python bench.py; PROF_LINES=30 bash tools/prof.sh s1no --no-overlap
# s2   pytest (ORB subset); python bench.py --no-cpu-baseline (x2); SLIDEO_RESIZE_GENERIC=1 python bench.py --no-cpu-baseline; tools/prof.sh
# s4   REPS=3 bash tools/ab_libs.sh s4 "" start=... new=product
# s5   pytest (ORB subset); REPS=3 bash tools/ab_libs.sh s5 "" fg0=<-DFAST_GROUPED=0> grouped=product
# s7   REPS=4 bash tools/ab_libs.sh s7 "--steps 300 --warmup 10" start=... fg0=... fw6=<-DFAST_WAVES_EU=6> new=product
# s8   pytest (ORB subset); REPS=4 bash tools/ab_libs.sh s8 "--steps 300 --warmup 10" prev=... fw6=... new=product        (FAST loader, 79 registers)
# s9   ... prev=... new=product                                                                                          (describe_blurred at 60 registers)
# s10  REPS=3 bash tools/ab_libs.sh s10 "--steps 300 --warmup 10" new=product ral=<-DRESIZE_ALIGNED>; tools/prof.sh for both
# s14  ... np0=<-DFAST_POOL_GROUPS=0> pooled=product
# s15  SLIDEO_LIB_PATH=<-DKT_PROBE> python bench.py --steps 40 --warmup 5 --no-cpu-baseline; SLIDEO_KNN_SHARE=1 ... --no-overlap; ... --no-overlap
# s17  pytest (kNN subset); REPS=4 bash tools/ab_libs.sh s17 ... prev=... new=product                                     (search slow path)
# s18  REPS=3 bash tools/ab_libs.sh s18 ... base=product n23=<-DKT_N23_EARLY> r5a3=<-DKT_RING_V=5 -DKT_AHEAD_V=3> r5a2=<-DKT_RING_V=5 -DKT_AHEAD_V=2>
# s20  python -m pytest tests -m gpu -x -q; bash tools/final_profile.sh r05fin      -> tools/collect_profiles.sh r05fin r05; tools/make_knn_traffic.py
# s21  REPS=3 bash tools/ab_libs.sh s21 ... base=product slots5=<-DSLIDEO_NSLOTS=5> slots6=<-DSLIDEO_NSLOTS=6>; python bench.py --unit 128|160|192|224|256
# s22  python bench.py --workload cfg3 --steps 20 --warmup 2 --no-cpu-baseline, with and without SLIDEO_KNN_SHARE=0
