#!/bin/bash
# round 3, visit 26: LDS ring depth of the weight-gradient launch, 3 against 4, alternating in one process
set -u
mkdir -p gpurun_out
timeout 600 python tools/ab_env.py --env RLX_DW_NBUF --values 3,4 --rounds 4 --steps 40 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_v26_ab_dw_nbuf.txt
