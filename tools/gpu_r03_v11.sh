#!/bin/bash
# round 3, visit 11: where the device idles inside the benchmark loop (gaps between consecutive kernels of the kernel trace)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r03_v11
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v11 -o bench -- python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v11_bench_prof.log 2>&1
DB=$(ls gpurun_out/prof_r03_v11/*.db gpurun_out/prof_r03_v11/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" --gaps > gpurun_out/r03_v11_bench_gaps.txt 2>&1; cat gpurun_out/r03_v11_bench_gaps.txt; fi
rm -rf gpurun_out/prof_r03_v11
tail -2 gpurun_out/r03_v11_bench_prof.log | cut -c1-300
