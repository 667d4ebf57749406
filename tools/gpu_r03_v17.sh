#!/bin/bash
# round 3, visit 17: pipeline-mode run-ahead A/B after the numpy staging copy
set -u
mkdir -p gpurun_out
for e in 1 4 2; do
echo "pipeline epochs $e" | tee -a gpurun_out/r03_v17_ab_defer_pipeline.txt
timeout 600 python tools/ab_defer.py --steps 30 --rounds 2 --pipeline-epochs $e 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_v17_ab_defer_pipeline.txt
done
