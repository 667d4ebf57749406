#!/bin/bash
# Round 2, GPU visit 4: the whole -m gpu suite (hard per-group timeouts), kernel trace, full bench line (reference-kind CPU baseline).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/e2e_bench_config_parity.jsonl
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -4 gpurun_out/$name.log | cut -c1-300; }
run v4_t_dist 420 python -m pytest tests/test_distributed.py -q -m gpu
run v4_t_new 600 python -m pytest tests/test_gpu_ext_real_registry.py tests/test_reference_entry_point.py tests/test_gpu_advantages.py tests/test_gpu_losses.py tests/test_end_to_end.py -q -m gpu
run v4_t_rest 900 python -m pytest tests -q -m gpu --deselect tests/test_distributed.py --ignore=tests/test_gpu_ext_real_registry.py --ignore=tests/test_reference_entry_point.py --ignore=tests/test_gpu_advantages.py --ignore=tests/test_gpu_losses.py --ignore=tests/test_end_to_end.py
rm -rf gpurun_out/prof_bench_bf16
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench_bf16 -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --no-token-tier > gpurun_out/v4_bench_prof.log 2>&1
echo "prof rc=$?"; tail -1 gpurun_out/v4_bench_prof.log | cut -c1-200
DB=$(ls gpurun_out/prof_bench_bf16/*.db gpurun_out/prof_bench_bf16/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v4_bench_kernels.txt 2>&1; head -9 gpurun_out/v4_bench_kernels.txt; fi
run v4_bench 900 python bench.py --no-traffic
