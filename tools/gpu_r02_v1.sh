#!/bin/bash
# Round 2, GPU visit 1: new parity tests (bench configuration, sync no-auto-reset, K2 ties, syncer apply, 2-rank xGMI / graph),
# the xGMI IPC probe, bf16 phase stamps + PMC counters of the fused launches, a quick bench line + kernel trace.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/e2e_bench_config_parity.jsonl
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -4 gpurun_out/$name.log | cut -c1-300; }
run v1_xgmi_probe 240 python tools/xgmi_probe.py
run v1_t_dist 900 python -m pytest tests/test_distributed.py -q -m gpu
run v1_t_benchcfg 1200 python -m pytest tests/test_end_to_end_bench_config.py -q -m gpu -s
run v1_t_e2e 900 python -m pytest tests/test_end_to_end.py -q -m gpu
run v1_t_misc 900 python -m pytest tests/test_gpu_token_path.py tests/test_gpu_advantages.py tests/test_gpu_weight_bucket.py tests/test_gpu_losses.py tests/test_gpu_fused_step.py -q -m gpu
run v1_phase 300 python tools/phase_times.py 8192
bash tools/pmc_step.sh 8192 bf16 > gpurun_out/v1_pmc_bf16.txt 2>&1; tail -12 gpurun_out/v1_pmc_bf16.txt | cut -c1-400
rm -rf gpurun_out/prof_bench_bf16
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench_bf16 -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --no-token-tier > gpurun_out/v1_bench_prof.log 2>&1
echo "prof rc=$?"; tail -1 gpurun_out/v1_bench_prof.log | cut -c1-300
DB=$(ls gpurun_out/prof_bench_bf16/*.db gpurun_out/prof_bench_bf16/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v1_bench_kernels.txt 2>&1; head -14 gpurun_out/v1_bench_kernels.txt; fi
run v1_bench 600 python bench.py --no-cpu-baseline --no-traffic --no-token-tier
