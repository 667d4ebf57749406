#!/bin/bash
# round 3, visit 24: how long does the weight-gradient launch take when its operands are warm in the XCD's L2?  (timing only:
# RLX_DW_REPEAT=2 walks the k-blocks twice; duration(2) - duration(1) = the warm pass)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for r in 1 2 3; do
rm -rf gpurun_out/prof_r03_v24
RLX_DW_REPEAT=$r timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v24 -o bench -- python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v24_prof$r.log 2>&1
DB=$(ls gpurun_out/prof_r03_v24/*.db gpurun_out/prof_r03_v24/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then echo "RLX_DW_REPEAT=$r" | tee -a gpurun_out/r03_v24_dw_warm_pass.txt; python tools/rocpd_stats.py "$DB" 2>&1 | grep "ppo_step_dw\|ppo_step_fused\|grad_reduce" | tee -a gpurun_out/r03_v24_dw_warm_pass.txt; fi
done
rm -rf gpurun_out/prof_r03_v24
