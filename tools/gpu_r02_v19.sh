#!/bin/bash
# Round 2, GPU visit 19: verification pass -- full GPU suite, smoke, final bench lines (bf16 with cpu_baseline + roofline, f32, no graph, pipeline), kernel traces.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -3 gpurun_out/$name.log | cut -c1-250; }
run v19_t_all 1800 python -m pytest tests -q -m gpu -x
run v19_smoke 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
run v19_bench_a 900 python bench.py
run v19_bench_b 600 python bench.py --no-cpu-baseline --no-roofline
run v19_bench_f32 600 python bench.py --precision 32 --no-cpu-baseline --no-roofline
run v19_bench_nograph 600 python bench.py --no-graph --no-cpu-baseline --no-roofline
run v19_bench_pipe 600 python bench.py --pipeline --rollout-epochs 4 --no-cpu-baseline --no-roofline
for prec in bf16 32; do
rm -rf gpurun_out/prof_bench_$prec
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench_$prec -o bench -- python bench.py --precision $prec --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/v19_bench_prof_$prec.log 2>&1
DB=$(ls gpurun_out/prof_bench_$prec/*.db gpurun_out/prof_bench_$prec/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v19_bench_kernels_$prec.txt 2>&1; head -12 gpurun_out/v19_bench_kernels_$prec.txt; fi
done
