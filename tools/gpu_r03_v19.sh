#!/bin/bash
# round 3, visit 19 (re-run as 20 with the ballot-sliced scans): is the codec's measure launch bound by the workgroup start rate?  grid-stride walk against one workgroup per piece
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_weight_patch.py -q -p no:cacheprovider 2>&1 | tail -2
for g in 0 2048; do
rm -rf gpurun_out/prof_r03_v20
RLX_ZPLANE_MEASURE_GRID=$g timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v20 -o zp -- python tools/bench_zplane.py > /dev/null 2>&1
DB=$(ls gpurun_out/prof_r03_v20/*.db gpurun_out/prof_r03_v20/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then echo "measure grid cap $g" | tee -a gpurun_out/r03_v20_ballot_scans.txt; python tools/rocpd_stats.py "$DB" 2>&1 | grep "zplane_" | tee -a gpurun_out/r03_v20_ballot_scans.txt; fi
done
rm -rf gpurun_out/prof_r03_v20
