#!/bin/bash
# Round 2, GPU visit 14: two 32-row workgroups per CU in the fused bf16 step (RLX_FUSED_RT=2) against one 64-row workgroup.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -3 gpurun_out/$name.log | cut -c1-250; }
RLX_FUSED_RT=2 run v14_t_rt2 900 python -m pytest tests -q -m gpu -x -k "fused_step or end_to_end"
for rt in 4 2 4 2; do
RLX_FUSED_RT=$rt timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > gpurun_out/v14_bench_rt$rt.log 2>&1
echo "rt=$rt rc=$? $(tail -1 gpurun_out/v14_bench_rt$rt.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"])' 2>&1 | tail -1)"
done
for rt in 2 4; do
rm -rf gpurun_out/prof_rt$rt
RLX_FUSED_RT=$rt timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rt$rt -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/v14_prof_rt$rt.log 2>&1
DB=$(ls gpurun_out/prof_rt$rt/*.db gpurun_out/prof_rt$rt/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v14_kernels_rt$rt.txt 2>&1; head -6 gpurun_out/v14_kernels_rt$rt.txt; fi
done
