#!/bin/bash
# Round 2, GPU visit 22: merged element + loss + dOut pass in the fused bf16 step (fast path), parity + bench + stamps.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -2 gpurun_out/$name.log | cut -c1-200; }
run v22_t_step 1200 python -m pytest tests -q -m gpu -x -k "fused_step or end_to_end or smoke or async or decoupled"
run v22_bench_a 600 python bench.py --no-cpu-baseline --no-roofline
run v22_bench_b 600 python bench.py --no-cpu-baseline --no-roofline
rm -rf gpurun_out/prof_bench_bf16
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench_bf16 -o bench -- python bench.py --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/v22_bench_prof_bf16.log 2>&1
DB=$(ls gpurun_out/prof_bench_bf16/*.db gpurun_out/prof_bench_bf16/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v22_bench_kernels_bf16.txt 2>&1; head -6 gpurun_out/v22_bench_kernels_bf16.txt; fi
run v22_phase 300 python tools/phase_times.py 8192
grep -A1 "fused_bf16" gpurun_out/v22_phase.log | cut -c1-700
