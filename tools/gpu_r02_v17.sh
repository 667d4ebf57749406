#!/bin/bash
# Round 2, GPU visit 17: does the LDS allocation size itself cost time?  fused 32-row tiles with padded LDS; dW with 2 / 3 buffers.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
RLX_DW_NBUF=2 timeout 600 python -m pytest tests/test_gpu_fused_step.py -q -m gpu -x -k "bf16" > gpurun_out/v17_t_nb2.log 2>&1; echo "t_nb2 rc=$?"; tail -1 gpurun_out/v17_t_nb2.log
for rep in 1 2; do
for cfg in "0 3" "8192 3" "20480 3" "0 2"; do
set -- $cfg
RLX_FUSED_LDS_PAD=$1 RLX_DW_NBUF=$2 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > gpurun_out/v17_bench_pad$1_nb$2.log 2>&1
echo "pad=$1 nbuf=$2 rc=$? $(tail -1 gpurun_out/v17_bench_pad$1_nb$2.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"])' 2>&1 | tail -1)"
done
done
for cfg in "20480 3" "0 2"; do
set -- $cfg
rm -rf gpurun_out/prof_p$1_$2
RLX_FUSED_LDS_PAD=$1 RLX_DW_NBUF=$2 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_p$1_$2 -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/v17_prof.log 2>&1
DB=$(ls gpurun_out/prof_p$1_$2/*.db gpurun_out/prof_p$1_$2/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v17_kernels_pad$1_nb$2.txt 2>&1; head -6 gpurun_out/v17_kernels_pad$1_nb$2.txt; fi
done
