#!/bin/bash
# Round 2, GPU visit 26: A/B on one box -- merged loss pass on / off, alternating; kernel trace of both.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2 3; do
for m in 1 0; do
RLX_FUSED_MERGED=$m timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > gpurun_out/v26_bench_m$m.log 2>&1
echo "merged=$m rc=$? $(tail -1 gpurun_out/v26_bench_m$m.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"])' 2>&1 | tail -1)"
done
done
for m in 1 0; do
rm -rf gpurun_out/prof_m$m
RLX_FUSED_MERGED=$m timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_m$m -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/v26_prof_m$m.log 2>&1
DB=$(ls gpurun_out/prof_m$m/*.db gpurun_out/prof_m$m/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v26_kernels_m$m.txt 2>&1; head -6 gpurun_out/v26_kernels_m$m.txt; fi
done
