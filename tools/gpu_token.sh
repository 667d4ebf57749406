#!/bin/bash
# Token-tier GPU visit: parity tests of the tier, the roofline probe (bf16 + f32), a kernel trace of the probe.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_token_path.py -m gpu -x -q > gpurun_out/pytest_token.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/pytest_token.log
timeout 600 python tools/bench_token.py --tokens 8192 --vocab 151936 --dtype bf16 > gpurun_out/token_bench_bf16.jsonl 2>gpurun_out/token_bench_bf16.err; echo "bench bf16 rc=$?"
cat gpurun_out/token_bench_bf16.jsonl; tail -3 gpurun_out/token_bench_bf16.err
timeout 600 python tools/bench_token.py --tokens 4096 --vocab 151936 --dtype f32 > gpurun_out/token_bench_f32.jsonl 2>gpurun_out/token_bench_f32.err; echo "bench f32 rc=$?"
cat gpurun_out/token_bench_f32.jsonl
timeout 600 python tools/bench_token.py --tokens 8192 --vocab 151936 --dtype bf16 --temperature 0.6 --mask-frac 0.6 > gpurun_out/token_bench_bf16_t06.jsonl 2>&1
cat gpurun_out/token_bench_bf16_t06.jsonl
rm -rf gpurun_out/prof_token
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_token -o tok -- python tools/bench_token.py --tokens 8192 --vocab 151936 --dtype bf16 --iters 10 > gpurun_out/token_prof.log 2>&1
DB=$(ls gpurun_out/prof_token/*.db gpurun_out/prof_token/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/token_kernels.txt 2>&1; head -12 gpurun_out/token_kernels.txt; fi
