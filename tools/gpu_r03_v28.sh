#!/bin/bash
# round 3, visit 28: the per-rank compute of an N-GPU strong-scaling job, measured on one GPU (section 6's table, compute rows)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for n in 2 4 8; do
rm -rf gpurun_out/prof_r03_v28
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v28 -o share -- python tools/per_rank_share.py --share $n --steps 20 > gpurun_out/r03_v28_share$n.log 2>&1
grep "share 1/" gpurun_out/r03_v28_share$n.log | tee -a gpurun_out/r03_v28_per_rank_share.txt
DB=$(ls gpurun_out/prof_r03_v28/*.db gpurun_out/prof_r03_v28/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" 2>&1 | head -7 | cut -c1-150 | tee -a gpurun_out/r03_v28_per_rank_share.txt; fi
done
rm -rf gpurun_out/prof_r03_v28
for n in 2 4 8; do timeout 200 python tools/per_rank_share.py --share $n --steps 40 2>&1 | grep "share 1/" | sed 's/^/without the tracer: /' | tee -a gpurun_out/r03_v28_per_rank_share.txt; done
