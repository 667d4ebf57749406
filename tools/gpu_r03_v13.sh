#!/bin/bash
# round 3, visit 13: A/B of the run-ahead loop inside one process
set -u
mkdir -p gpurun_out
timeout 600 python tools/ab_defer.py --steps 40 --rounds 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_v13_ab_defer.txt
