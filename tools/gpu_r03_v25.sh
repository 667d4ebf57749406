#!/bin/bash
# round 3, visit 25: per-k-block stamps of the weight-gradient launch, cold pass against the L2-warm second pass
set -u
mkdir -p gpurun_out
for r in 1 2; do
echo "RLX_DW_REPEAT=$r" | tee -a gpurun_out/r03_v25_dw_stamps.txt
RLX_DW_REPEAT=$r timeout 300 python tools/phase_times.py 8192 > gpurun_out/r03_v25_phase$r.log 2>&1; echo "rc=$?"
grep "dw bf16" gpurun_out/r03_v25_phase$r.log | tee -a gpurun_out/r03_v25_dw_stamps.txt
tail -3 gpurun_out/r03_v25_phase$r.log | cut -c1-300
done
