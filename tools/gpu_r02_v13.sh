#!/bin/bash
# Round 2, GPU visit 13: slab reduce with all slab loads in flight; optimizer / e2e tests; slab-count sweep; trace.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -3 gpurun_out/$name.log | cut -c1-250; }
run v13_t_opt 900 python -m pytest tests -q -m gpu -x -k "adamw or optimizer or end_to_end or fused_step or distributed or xgmi"
for s in 24 20 16 12; do
RLX_DW_SLABS=$s timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > gpurun_out/v13_bench_s$s.log 2>&1
echo "slabs=$s rc=$? $(tail -1 gpurun_out/v13_bench_s$s.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"])' 2>&1 | tail -1)"
done
for s in 24 16; do
rm -rf gpurun_out/prof_s$s
RLX_DW_SLABS=$s timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s$s -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/v13_prof_s$s.log 2>&1
DB=$(ls gpurun_out/prof_s$s/*.db gpurun_out/prof_s$s/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v13_kernels_s$s.txt 2>&1; head -6 gpurun_out/v13_kernels_s$s.txt; fi
done
