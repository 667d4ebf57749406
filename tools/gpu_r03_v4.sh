#!/bin/bash
# round 3, visit 4: codec v2 (word-level plane transposes) parity + throughput, the N > 1 bench path after the watchdog change,
# stall counters of the mixed read/write streams
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_weight_patch.py tests/test_distributed.py tests/test_gpu_reasoning_loop.py -q -p no:cacheprovider > gpurun_out/r03_v4_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_v4_pytest.log
grep -n "passed\|failed\|error" gpurun_out/r03_v4_pytest.log | tail -4
timeout 300 python tools/bench_zplane.py > gpurun_out/r03_v4_zplane.jsonl 2>&1; cut -c1-420 gpurun_out/r03_v4_zplane.jsonl
bash tools/pmc_stalls.sh > gpurun_out/r03_v4_pmc_stalls.log 2>&1; tail -30 gpurun_out/pmc_stalls.txt
