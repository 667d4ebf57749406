#!/bin/bash
# round 3, visit 5: the early-hand-off segmented GAE scan against the streaming scan in the HBM regime (bit-exactness + time)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/sweep_gae_hbm.py 65536 128 > gpurun_out/r03_v5_gae_handoff_sweep.txt 2>&1; grep -v "^rep0" gpurun_out/r03_v5_gae_handoff_sweep.txt | tail -40
timeout 300 python tools/sweep_gae_hbm.py 262144 128 2>&1 | grep "rep1" | grep "stream rows=64 nt=1\|handoff" > gpurun_out/r03_v5_gae_handoff_sweep_262144.txt; cat gpurun_out/r03_v5_gae_handoff_sweep_262144.txt
