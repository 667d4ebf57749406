#!/bin/bash
# Round 2, GPU visit 27: what the tile scatter costs inside the AdamW launch (timing only) + fused-step tests after the launch tidy-up.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -k "fused_step or optimizer or smoke" > gpurun_out/v27_t.log 2>&1; echo "t rc=$?"; tail -1 gpurun_out/v27_t.log
for skip in 0 1; do
rm -rf gpurun_out/prof_sk$skip
if [ $skip = 1 ]; then export RLX_ADAMW_SKIP_TILES=1; fi
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sk$skip -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/v27_prof_sk$skip.log 2>&1
DB=$(ls gpurun_out/prof_sk$skip/*.db gpurun_out/prof_sk$skip/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/v27_kernels_sk$skip.txt 2>&1; head -6 gpurun_out/v27_kernels_sk$skip.txt; fi
done
