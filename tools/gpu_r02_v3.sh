#!/bin/bash
# Round 2, GPU visit 3: 2-rank transports again (longer wait bound), bf16 numerics after the packed-tanh / staged-loss-input fused
# kernel, stamps, kernel traces (with and without the optimizer's tile scatter).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/e2e_bench_config_parity.jsonl
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -4 gpurun_out/$name.log | cut -c1-300; }
run v3_t_dist 900 python -m pytest tests/test_distributed.py -q -m gpu
run v3_t_fused 900 python -m pytest tests/test_gpu_fused_step.py tests/test_policy.py tests/test_gpu_losses.py -q -m gpu
run v3_t_benchcfg 1200 python -m pytest tests/test_end_to_end_bench_config.py -q -m gpu -s
run v3_t_e2e 900 python -m pytest tests/test_end_to_end.py tests/test_reference_entry_point.py tests/test_gpu_weight_bucket.py -q -m gpu
run v3_phase 300 python tools/phase_times.py 8192
for tag in tiles notiles; do
rm -rf gpurun_out/prof_bench_$tag
if [ $tag = notiles ]; then export RLX_BENCH_OPT_TILES=0; else export RLX_BENCH_OPT_TILES=1; fi
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench_$tag -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --no-token-tier > gpurun_out/v3_bench_prof_$tag.log 2>&1
echo "prof $tag rc=$?"; tail -1 gpurun_out/v3_bench_prof_$tag.log | cut -c1-200
DB=$(ls gpurun_out/prof_bench_$tag/*.db gpurun_out/prof_bench_$tag/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v3_bench_kernels_$tag.txt 2>&1; head -9 gpurun_out/v3_bench_kernels_$tag.txt; fi
done
export RLX_BENCH_OPT_TILES=1
run v3_bench 600 python bench.py --no-cpu-baseline --no-traffic --no-token-tier
