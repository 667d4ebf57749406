#!/bin/bash
# round 3, visit 2: the tests that failed / are new (codec, compressed transport, host-snapshot builder, in-process xGMI groups, bf16
# C = 2 fused step), a kernel trace of the bench loop, pipeline-mode lines after the randperm fix, the codec's roofline probe
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_weight_patch.py "tests/test_distributed.py::test_xgmi_exchange_in_process_group" "tests/test_gpu_fused_step.py::test_ppo_step_two_action_chunks" -q -p no:cacheprovider > gpurun_out/r03_v2_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_v2_pytest.log
grep -n "passed\|failed\|error" gpurun_out/r03_v2_pytest.log | tail -5
rm -rf gpurun_out/prof_r03_v2
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v2 -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v2_bench_prof.log 2>&1
echo "prof rc=$?"; tail -c 400 gpurun_out/r03_v2_bench_prof.log
DB=$(ls gpurun_out/prof_r03_v2/*.db gpurun_out/prof_r03_v2/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/r03_v2_bench_kernels_bf16.txt 2>&1; head -14 gpurun_out/r03_v2_bench_kernels_bf16.txt; fi
rm -rf gpurun_out/prof_r03_v2
for e in 1 4; do
timeout 300 python bench.py --pipeline --rollout-epochs $e --steps 50 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v2_pipeline_e$e.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r03_v2_pipeline_e$e.json').read().strip().splitlines()[-1]);print('pipeline e$e', d['ms_per_step'], d['ms_per_step_windows'])"
done
timeout 300 python bench.py --steps 50 --no-cpu-baseline --no-roofline --no-variants 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('sync', d['ms_per_step'], d['ms_per_step_windows'])"
timeout 300 python tools/bench_zplane.py > gpurun_out/r03_v2_zplane.jsonl 2>&1; cat gpurun_out/r03_v2_zplane.jsonl | cut -c1-330
