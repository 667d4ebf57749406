#!/bin/bash
# Round 2, GPU visit 16: LDS ring depth of the bf16 weight-gradient launch x split-K slab count.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -3 gpurun_out/$name.log | cut -c1-250; }
RLX_DW_NBUF=9 run v16_t_nb9 600 python -m pytest tests/test_gpu_fused_step.py -q -m gpu -x -k "bf16"
RLX_DW_NBUF=4 run v16_t_nb4 600 python -m pytest tests/test_gpu_fused_step.py -q -m gpu -x -k "bf16"
run v16_t_dist 300 python -m pytest tests/test_distributed.py -q -m gpu -x
for cfg in "3 16" "4 16" "4 24" "5 16" "6 12" "6 16" "9 12" "9 8" "9 16" "3 16"; do
set -- $cfg
RLX_DW_NBUF=$1 RLX_DW_SLABS=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > gpurun_out/v16_bench_nb$1_s$2.log 2>&1
echo "nbuf=$1 slabs=$2 rc=$? $(tail -1 gpurun_out/v16_bench_nb$1_s$2.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"])' 2>&1 | tail -1)"
done
for cfg in "9 12" "6 16"; do
set -- $cfg
rm -rf gpurun_out/prof_nb$1
RLX_DW_NBUF=$1 RLX_DW_SLABS=$2 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_nb$1 -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/v16_prof_nb$1.log 2>&1
DB=$(ls gpurun_out/prof_nb$1/*.db gpurun_out/prof_nb$1/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v16_kernels_nb$1_s$2.txt 2>&1; head -6 gpurun_out/v16_kernels_nb$1_s$2.txt; fi
done
