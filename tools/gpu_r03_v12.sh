#!/bin/bash
# round 3, visit 12: the run-ahead loop (metrics read one iteration late) -- equivalence tests, then the headline with and
# without it on the same box, and the idle gaps of the kernel trace with it
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_end_to_end.py tests/test_end_to_end_bench_config.py tests/test_distributed.py -m gpu -q -p no:cacheprovider > gpurun_out/r03_v12_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_v12_pytest.log
grep -n "passed\|failed\|error\|Error" gpurun_out/r03_v12_pytest.log | tail -12
for d in 0 1 0 1; do
RLX_BENCH_DEFER_METRICS=$d timeout 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v12_defer$d.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r03_v12_defer$d.json').read().strip().splitlines()[-1]);print('defer=$d', d['ms_per_step'], d['ms_per_step_windows'], d['value'])"
done
rm -rf gpurun_out/prof_r03_v12
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v12 -o bench -- python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v12_bench_prof.log 2>&1
DB=$(ls gpurun_out/prof_r03_v12/*.db gpurun_out/prof_r03_v12/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" --gaps > gpurun_out/r03_v12_bench_gaps.txt 2>&1; cat gpurun_out/r03_v12_bench_gaps.txt; fi
rm -rf gpurun_out/prof_r03_v12
