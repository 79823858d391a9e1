#!/bin/bash
# round 6, run 23: the deck-size rule for the search's block shape (knn_w12_ratio) as the default: configs[3], configs[4], headline, 800 pages; the modes test
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run23; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "blocks_per_cu" > $out/tests.log 2>&1; tail -2 $out/tests.log
REPS=2 tools/ab_env.sh r06_rule_cfg3 "--workload cfg3 --total-frames 20480 --steps 8 --warmup 2 --no-host-frames" default="" never="SLIDEO_KNN_W12_RATIO=0"
REPS=2 tools/ab_env.sh r06_rule_cfg4 "--workload cfg4 --steps 12 --warmup 4 --no-host-frames" default="" never="SLIDEO_KNN_W12_RATIO=0"
REPS=2 tools/ab_env.sh r06_rule_head "--steps 40 --warmup 4 --no-host-frames" default="" never="SLIDEO_KNN_W12_RATIO=0"
REPS=1 tools/ab_env.sh r06_rule_p800 "--pages 800 --steps 40 --warmup 4 --no-host-frames" default="" never="SLIDEO_KNN_W12_RATIO=0"
python bench.py --workload cfg3 --steps 20 --warmup 2 --no-cpu-baseline --no-host-frames 2>/dev/null | tail -1 > $out/cfg3_full.json
python -c "import json;j=json.load(open('$out/cfg3_full.json'));print('cfg3 full job', j['value'], j['ms_per_step'], j['config']['lecture'])"
