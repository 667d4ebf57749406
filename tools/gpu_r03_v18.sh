#!/bin/bash
# round 3, visit 18: single-pass patch encoder (decoupled look-back) against the multi-launch one
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for sp in 1 0; do
RLX_ZPLANE_SINGLE_PASS=$sp timeout 600 python -m pytest tests/test_gpu_weight_patch.py -q -p no:cacheprovider > gpurun_out/r03_v18_pytest_sp$sp.log 2>&1; echo "single_pass=$sp pytest rc=$?"
grep -n "passed\|failed\|error\|Error" gpurun_out/r03_v18_pytest_sp$sp.log | tail -5
RLX_ZPLANE_SINGLE_PASS=$sp timeout 300 python tools/bench_zplane.py > gpurun_out/r03_v18_zplane_sp$sp.jsonl 2>&1
python - <<PY
import json
for ln in open('gpurun_out/r03_v18_zplane_sp$sp.jsonl'):
    if ln.startswith('{'):
        d=json.loads(ln); print(d['stream'][:28], d['ratio'], d['round_trip_ok'], 'c', d['compress_us'], d['compress_frac_of_8TBps'], 'd', d['decompress_us'], d['decompress_frac_of_8TBps'])
PY
done
rm -rf gpurun_out/prof_r03_v18
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v18 -o zp -- python tools/bench_zplane.py > /dev/null 2>&1
DB=$(ls gpurun_out/prof_r03_v18/*.db gpurun_out/prof_r03_v18/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/r03_v18_zplane_kernels.txt 2>&1; grep -i "zplane\|kernel " gpurun_out/r03_v18_zplane_kernels.txt | head -12; fi
rm -rf gpurun_out/prof_r03_v18
