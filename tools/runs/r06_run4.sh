#!/bin/bash
# round 6, run 4: MFMA chain probe with the 'split' schedule; search traffic at the configs[3] / configs[4] deck sizes
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_run4; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain_probe tools/mfma_chain_probe.hip > /dev/null 2>&1 && /tmp/mfma_chain_probe > $out/mfma_chain_probe.txt; cat $out/mfma_chain_probe.txt
bash tools/pmc_traffic_decks.sh r06 > $out/traffic.log 2>&1; cat $out/traffic.log
