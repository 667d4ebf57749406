#!/bin/bash
# round 3, visit 22: pipeline mode with four overlapped rollout epochs -- kernel durations and idle gaps (why is it not at
# update-phase + one rollout epoch?)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r03_v22
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v22 -o bench -- python bench.py --gpus 1 --steps 30 --warmup 5 --pipeline --rollout-epochs 4 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v22_bench_prof.log 2>&1
DB=$(ls gpurun_out/prof_r03_v22/*.db gpurun_out/prof_r03_v22/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/r03_v22_pipeline_e4_kernels.txt 2>&1; head -24 gpurun_out/r03_v22_pipeline_e4_kernels.txt | cut -c1-150; python tools/rocpd_stats.py "$DB" --gaps > gpurun_out/r03_v22_pipeline_e4_gaps.txt 2>&1; cat gpurun_out/r03_v22_pipeline_e4_gaps.txt | cut -c1-190; fi
rm -rf gpurun_out/prof_r03_v22
tail -1 gpurun_out/r03_v22_bench_prof.log | cut -c1-400
