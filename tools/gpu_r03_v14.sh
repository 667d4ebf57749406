#!/bin/bash
# round 3, visit 14: XCD-contiguous row tiles in the fused launch (every XCD writes the rows the weight-gradient launch makes it read)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for x in 0 1; do
rm -rf gpurun_out/prof_r03_v14
RLX_FUSED_XCD_ROWS=$x timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v14 -o bench -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v14_bench_prof$x.log 2>&1
DB=$(ls gpurun_out/prof_r03_v14/*.db gpurun_out/prof_r03_v14/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/r03_v14_kernels_xcd$x.txt 2>&1; echo "xcd_rows=$x"; head -6 gpurun_out/r03_v14_kernels_xcd$x.txt; fi
done
rm -rf gpurun_out/prof_r03_v14
timeout 900 python -m pytest tests/test_gpu_fused_step.py tests/test_end_to_end_bench_config.py -q -p no:cacheprovider > gpurun_out/r03_v14_pytest.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|error\|Error" gpurun_out/r03_v14_pytest.log | tail -6
