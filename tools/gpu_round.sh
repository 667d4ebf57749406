#!/bin/bash
# One GPU-box visit: parity tests, a kernel-trace profile of the bench, and the plain bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
if [ "${1:-}" = "tests" ]; then exit 0; fi
rm -rf gpurun_out/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
echo "prof rc=$?"
tail -2 gpurun_out/bench_prof.log
DB=$(ls gpurun_out/prof_bench/*.db gpurun_out/prof_bench/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/bench_kernels.txt 2>&1; head -30 gpurun_out/bench_kernels.txt; fi
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log
timeout 600 python bench.py --no-graph --no-cpu-baseline --no-roofline > gpurun_out/bench_nograph.log 2>&1; tail -1 gpurun_out/bench_nograph.log
