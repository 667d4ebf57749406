#!/bin/bash
# One GPU-box visit: parity tests, smoke, kernel-trace profiles of the bench (both precisions), and the plain bench lines.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
grep -n "passed\|failed\|Error" gpurun_out/pytest_gpu.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
if [ "${1:-}" = "tests" ]; then exit 0; fi
for prec in bf16 32; do
rm -rf gpurun_out/prof_bench_$prec
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench_$prec -o bench -- python bench.py --precision $prec --steps 3 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/bench_prof_$prec.log 2>&1
echo "prof $prec rc=$?"; tail -1 gpurun_out/bench_prof_$prec.log | cut -c1-200
DB=$(ls gpurun_out/prof_bench_$prec/*.db gpurun_out/prof_bench_$prec/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/bench_kernels_$prec.txt 2>&1; head -12 gpurun_out/bench_kernels_$prec.txt; fi
done
timeout 600 python tools/bench_token.py --tokens 8192 --vocab 151936 --dtype bf16 > gpurun_out/token_bench_bf16.jsonl 2>/dev/null; cat gpurun_out/token_bench_bf16.jsonl
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log
timeout 600 python bench.py --precision 32 --no-cpu-baseline --no-roofline > gpurun_out/bench_f32.log 2>&1; tail -1 gpurun_out/bench_f32.log | cut -c1-300
timeout 600 python bench.py --no-graph --no-cpu-baseline --no-roofline > gpurun_out/bench_nograph.log 2>&1; tail -1 gpurun_out/bench_nograph.log | cut -c1-300
timeout 300 python tools/bench_widening.py > gpurun_out/widening_bench.jsonl 2>/dev/null; cat gpurun_out/widening_bench.jsonl
timeout 300 python tools/bench_bucket.py > gpurun_out/bucket_bench.jsonl 2>/dev/null; tail -4 gpurun_out/bucket_bench.jsonl
