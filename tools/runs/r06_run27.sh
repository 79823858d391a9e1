#!/bin/bash
# round 6, run 27: the search's matrix instruction WITHOUT the scale load (-DKT_UNSCALED: scale operands 0 select v_mfma_f32_32x32x64_f8f6f4,
# 64-bit encoding, no v_mfma_ld_scale): same function? (bitwise check), the chain probe both ways, parity on the variant library, A/B
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run27; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -o /tmp/chk tools/mfma_unscaled_check.hip 2>/dev/null && /tmp/chk | tee $out/check.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_s tools/mfma_chain_probe.hip 2>/dev/null && /tmp/probe_s > $out/probe_scaled.txt
hipcc --offload-arch=gfx950 -O3 -DMF_UNSCALED -o /tmp/probe_u tools/mfma_chain_probe.hip 2>/dev/null && /tmp/probe_u > $out/probe_unscaled.txt
paste -d'|' $out/probe_scaled.txt $out/probe_unscaled.txt | cut -c1-200
V=slideo_amd/lib/variants/unscaled/libslideo_amd.so
SLIDEO_LIB_PATH=$V timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn or end_to_end or dedup or fused or ratio or modes" > $out/parity.log 2>&1; tail -2 $out/parity.log
REPS=3 tools/ab_env.sh r06_unscaled "--steps 60 --no-host-frames" base="" unscaled="SLIDEO_LIB_PATH=$V"
REPS=1 tools/ab_env.sh r06_unscaled_cfg3 "--workload cfg3 --total-frames 20480 --steps 8 --warmup 2 --no-host-frames" base="" unscaled="SLIDEO_LIB_PATH=$V"
