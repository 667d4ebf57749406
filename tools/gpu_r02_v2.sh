#!/bin/bash
# Round 2, GPU visit 2: re-run what failed in visit 1 + the dW head-reduce / AdamW scalar fixes (kernel trace of the bench).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/e2e_bench_config_parity.jsonl
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -4 gpurun_out/$name.log | cut -c1-300; }
run v2_t_dist 900 python -m pytest tests/test_distributed.py -q -m gpu
run v2_t_benchcfg 1200 python -m pytest tests/test_end_to_end_bench_config.py -q -m gpu -s
run v2_t_misc 900 python -m pytest tests/test_gpu_weight_bucket.py tests/test_gpu_losses.py tests/test_gpu_fused_step.py tests/test_end_to_end.py -q -m gpu
run v2_phase 300 python tools/phase_times.py 8192
rm -rf gpurun_out/prof_bench_bf16
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench_bf16 -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --no-token-tier > gpurun_out/v2_bench_prof.log 2>&1
echo "prof rc=$?"; tail -1 gpurun_out/v2_bench_prof.log | cut -c1-200
DB=$(ls gpurun_out/prof_bench_bf16/*.db gpurun_out/prof_bench_bf16/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v2_bench_kernels.txt 2>&1; head -9 gpurun_out/v2_bench_kernels.txt; fi
run v2_bench 600 python bench.py --no-cpu-baseline --no-traffic --no-token-tier
run v2_bench_f32 600 python bench.py --precision 32 --no-cpu-baseline --no-roofline
