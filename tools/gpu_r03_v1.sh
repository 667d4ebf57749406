#!/bin/bash
# round 3, visit 1: full GPU suite (no -x: every failure of the new N>1 / C=2 / 10-iteration tests in one visit), smoke, the
# bench line as the driver calls it, the two-process xGMI probe (both forms, both hand-shakes)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/r03_v1_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_v1_pytest.log
grep -n "passed\|failed\|error" gpurun_out/r03_v1_pytest.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_v1_bench.log 2> gpurun_out/r03_v1_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r03_v1_bench.log
timeout 400 python tools/xgmi_probe.py > gpurun_out/r03_v1_xgmi_probe.log 2>&1; echo "probe rc=$?"; tail -12 gpurun_out/r03_v1_xgmi_probe.log
