#!/bin/bash
# round 3, visit 6: the f32 weight-gradient launch as 3 x bf16 splits (parity + time against the exact-f32 MFMA), the GAE hand-off
# variant's bit-exactness tests
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_advantages.py tests/test_end_to_end_bench_config.py tests/test_end_to_end.py -q -p no:cacheprovider > gpurun_out/r03_v6_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_v6_pytest.log
grep -n "passed\|failed\|error" gpurun_out/r03_v6_pytest.log | tail -4
for ex in 0 1; do
RLX_F32_EXACT_MFMA=$ex timeout 300 python bench.py --precision 32 --steps 40 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v6_f32_exact$ex.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r03_v6_f32_exact$ex.json').read().strip().splitlines()[-1]);print('f32 exact_mfma=$ex', d['ms_per_step'], d['ms_per_step_windows'])"
done
rm -rf gpurun_out/prof_r03_v6
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v6 -o bench -- python bench.py --precision 32 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v6_bench_prof.log 2>&1
DB=$(ls gpurun_out/prof_r03_v6/*.db gpurun_out/prof_r03_v6/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/r03_v6_bench_kernels_f32.txt 2>&1; head -8 gpurun_out/r03_v6_bench_kernels_f32.txt; fi
rm -rf gpurun_out/prof_r03_v6
