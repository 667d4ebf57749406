#!/bin/bash
# Round 2, GPU visit 15: 32-row fused tiles as default -- full GPU suite, PD variants, kernel trace, stamps, bench lines.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -3 gpurun_out/$name.log | cut -c1-250; }
run v15_t_all 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_distributed.py
run v15_t_dist 600 python -m pytest tests/test_distributed.py -q -m gpu -x
for pd in 4 3 2; do
RLX_FUSED_PD=$pd timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > gpurun_out/v15_bench_pd$pd.log 2>&1
echo "pd=$pd rc=$? $(tail -1 gpurun_out/v15_bench_pd$pd.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"])' 2>&1 | tail -1)"
done
rm -rf gpurun_out/prof_bench_bf16
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench_bf16 -o bench -- python bench.py --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/v15_bench_prof_bf16.log 2>&1
DB=$(ls gpurun_out/prof_bench_bf16/*.db gpurun_out/prof_bench_bf16/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v15_bench_kernels_bf16.txt 2>&1; head -8 gpurun_out/v15_bench_kernels_bf16.txt; fi
run v15_phase 300 python tools/phase_times.py 8192
grep -A1 "fused_bf16" gpurun_out/v15_phase.log | cut -c1-700
