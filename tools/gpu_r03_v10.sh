#!/bin/bash
# round 3, visit 8 (and 10, after the two-level scan + several byte-stream blocks per workgroup): the codec at cell granularity (register-level pack / unpack) -- parity tests, then its roofline probe; the
# reasoning learner's pipeline mode against the oracle loop
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_weight_patch.py tests/test_gpu_reasoning_loop.py -q -p no:cacheprovider > gpurun_out/r03_v10_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_v10_pytest.log
grep -n "passed\|failed\|error\|Error" gpurun_out/r03_v10_pytest.log | tail -12
timeout 300 python tools/bench_zplane.py > gpurun_out/r03_v10_zplane.jsonl 2>&1; echo "zplane rc=$?"
python - <<'PY'
import json
for ln in open('gpurun_out/r03_v10_zplane.jsonl'):
    if ln.startswith('{'):
        d=json.loads(ln); print(d['stream'][:28], d['ratio'], d['round_trip_ok'], 'c', d['compress_us'], d['compress_frac_of_8TBps'], 'd', d['decompress_us'], d['decompress_frac_of_8TBps'])
    else:
        print(ln.strip()[:200])
PY
rm -rf gpurun_out/prof_r03_v10
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v10 -o zp -- python tools/bench_zplane.py > /dev/null 2>&1
DB=$(ls gpurun_out/prof_r03_v10/*.db gpurun_out/prof_r03_v10/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/r03_v10_zplane_kernels.txt 2>&1; grep -i "zplane\|kernel " gpurun_out/r03_v10_zplane_kernels.txt | head -20; fi
rm -rf gpurun_out/prof_r03_v10
