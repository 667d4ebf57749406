#!/bin/bash
# round 3, visit 9: per-kernel times of the codec probe
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r03_v9
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v9 -o zp -- python tools/bench_zplane.py > gpurun_out/r03_v9_zplane.jsonl 2>&1
DB=$(ls gpurun_out/prof_r03_v9/*.db gpurun_out/prof_r03_v9/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/r03_v9_zplane_kernels.txt 2>&1; grep -i "zplane\|kernel " gpurun_out/r03_v9_zplane_kernels.txt | head -20; fi
rm -rf gpurun_out/prof_r03_v9
