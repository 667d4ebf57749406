#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_step
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_step -o step -- python tools/bench_step.py ${1:-8192} > gpurun_out/bench_step_prof.log 2>&1
echo "prof rc=$?"; grep "us per" gpurun_out/bench_step_prof.log
DB=$(ls gpurun_out/prof_step/*.db gpurun_out/prof_step/*/*.db 2>/dev/null | head -1)
python tools/rocpd_stats.py "$DB" rlx --by-grid > gpurun_out/step_kernels.txt 2>&1; cat gpurun_out/step_kernels.txt
