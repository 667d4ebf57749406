#!/bin/bash
# Round 2, GPU visit 18: rollout with 32-row tiles; bf16 PMC counters of the final fused / dW launches.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
RLX_ROLLOUT_RT=2 timeout 900 python -m pytest tests -q -m gpu -x -k "rollout or end_to_end" > gpurun_out/v18_t_rt2.log 2>&1; echo "t_rollout_rt2 rc=$?"; tail -1 gpurun_out/v18_t_rt2.log
for rep in 1 2; do
for rt in 1 2; do
RLX_ROLLOUT_RT=$rt timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > gpurun_out/v18_bench_rrt$rt.log 2>&1
echo "rollout_rt=$rt rc=$? $(tail -1 gpurun_out/v18_bench_rrt$rt.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"])' 2>&1 | tail -1)"
done
done
rm -rf gpurun_out/prof_rrt2
RLX_ROLLOUT_RT=2 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rrt2 -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/v18_prof.log 2>&1
DB=$(ls gpurun_out/prof_rrt2/*.db gpurun_out/prof_rrt2/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v18_kernels_rrt2.txt 2>&1; head -7 gpurun_out/v18_kernels_rrt2.txt; fi
bash tools/pmc_step.sh 8192 bf16 > gpurun_out/v18_pmc_bf16.txt 2>&1; tail -30 gpurun_out/v18_pmc_bf16.txt | cut -c1-400
