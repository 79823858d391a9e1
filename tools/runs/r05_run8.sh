#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
for vr in 0 1; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --verify-model 1 --verdict-rule $vr > $O/headline_h_vr$vr.json 2> $O/headline_h_vr$vr.err
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --verify-model 1 --persp 0.1 --verdict-rule $vr > $O/headline_h_persp_vr$vr.json 2> $O/headline_h_persp_vr$vr.err
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --verify-model 0 --verdict-rule $vr > $O/headline_s_vr$vr.json 2> $O/headline_s_vr$vr.err
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --verify-model 0 --persp 0.1 --verdict-rule $vr > $O/headline_s_persp_vr$vr.json 2> $O/headline_s_persp_vr$vr.err
done
timeout 900 python bench.py --workload cfg4 --steps 20 --warmup 3 --no-cpu-baseline > $O/cfg4.json 2> $O/cfg4.err
timeout 900 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline --hdlt 0 > $O/cfg4_hdlt0.json 2> $O/cfg4_hdlt0.err
timeout 900 python bench.py --workload cfg4 --steps 20 --warmup 3 --no-cpu-baseline --verdict-rule 1 > $O/cfg4_vr1.json 2> $O/cfg4_vr1.err
timeout 1200 python bench.py --workload cfg3 --steps 20 --warmup 2 --no-cpu-baseline > $O/cfg3.json 2> $O/cfg3.err
timeout 900 python bench.py --workload cfg1 --steps 40 --warmup 5 --no-cpu-baseline > $O/cfg1.json 2> $O/cfg1.err
for f in $O/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    j=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][0]
    c=j['config']
    print(j['value'], j['ms_per_step'], 'acc', c['accuracy_vs_synthetic_truth'], 'vm', c['verify_model'], 'hdlt', c['ocv_hdlt'], 'persp', c['frames_projective_component'], c.get('lecture'))
except Exception as e: print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
