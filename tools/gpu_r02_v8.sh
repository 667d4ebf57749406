#!/bin/bash
# Round 2, GPU visit 8: AdamW transposed tile scatter -- optimizer / e2e tests, kernel traces (bf16 / f32), bench lines.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -3 gpurun_out/$name.log | cut -c1-250; }
run v8_t_opt 900 python -m pytest tests -q -m gpu -x -k "adamw or optimizer or end_to_end or ppo_step or weight_bucket"
for prec in bf16 32; do
rm -rf gpurun_out/prof_bench_$prec
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench_$prec -o bench -- python bench.py --precision $prec --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/v8_bench_prof_$prec.log 2>&1
echo "prof $prec rc=$?"; tail -1 gpurun_out/v8_bench_prof_$prec.log | cut -c1-160
DB=$(ls gpurun_out/prof_bench_$prec/*.db gpurun_out/prof_bench_$prec/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v8_bench_kernels_$prec.txt 2>&1; head -12 gpurun_out/v8_bench_kernels_$prec.txt; fi
done
run v8_bench_b 600 python bench.py --no-cpu-baseline --no-roofline
run v8_bench_f32 600 python bench.py --precision 32 --no-cpu-baseline --no-roofline
