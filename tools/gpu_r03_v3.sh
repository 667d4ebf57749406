#!/bin/bash
# round 3, visit 3: full GPU suite, counters of the fused step after the 32-bit slab stores, kernel trace, pipeline vs sync lines
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/r03_v3_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_v3_pytest.log
grep -n "passed\|failed\|error" gpurun_out/r03_v3_pytest.log | tail -4
bash tools/pmc_step.sh 8192 bf16 > gpurun_out/r03_v3_fused_step_pmc_bf16.txt 2>&1; grep "ppo_step_fused_bf16" gpurun_out/r03_v3_fused_step_pmc_bf16.txt | grep "LDS_BANK" | cut -c1-400
rm -rf gpurun_out/pmc
rm -rf gpurun_out/prof_r03_v3
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v3 -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v3_bench_prof.log 2>&1
DB=$(ls gpurun_out/prof_r03_v3/*.db gpurun_out/prof_r03_v3/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/r03_v3_bench_kernels_bf16.txt 2>&1; head -8 gpurun_out/r03_v3_bench_kernels_bf16.txt; fi
rm -rf gpurun_out/prof_r03_v3
for e in 1 4; do
timeout 300 python bench.py --pipeline --rollout-epochs $e --steps 100 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v3_pipeline_e$e.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r03_v3_pipeline_e$e.json').read().strip().splitlines()[-1]);print('pipeline e$e', d['ms_per_step'], d['ms_per_step_windows'])"
done
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v3_sync.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r03_v3_sync.json').read().strip().splitlines()[-1]);print('sync', d['ms_per_step'], d['ms_per_step_windows'])"
