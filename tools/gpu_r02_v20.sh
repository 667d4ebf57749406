#!/bin/bash
# Round 2, GPU visit 20: precision-dependent slab count (f32: 24, bf16: 16): step tests, f32 / bf16 bench lines, f32 trace.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -2 gpurun_out/$name.log | cut -c1-200; }
run v20_t_step 900 python -m pytest tests -q -m gpu -x -k "fused_step or end_to_end or optimizer or adamw"
run v20_smoke 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
run v20_bench_f32_a 600 python bench.py --precision 32 --no-cpu-baseline --no-roofline
run v20_bench_bf16 600 python bench.py --no-cpu-baseline --no-roofline
run v20_bench_f32_b 600 python bench.py --precision 32 --no-cpu-baseline --no-roofline
rm -rf gpurun_out/prof_bench_32
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench_32 -o bench -- python bench.py --precision 32 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/v20_bench_prof_32.log 2>&1
DB=$(ls gpurun_out/prof_bench_32/*.db gpurun_out/prof_bench_32/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v20_bench_kernels_32.txt 2>&1; head -7 gpurun_out/v20_bench_kernels_32.txt; fi
