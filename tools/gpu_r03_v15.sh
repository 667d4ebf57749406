#!/bin/bash
# round 3, visit 7 (re-run as visit 15 after the run-ahead loop and the codec rewrite): the full GPU suite, smoke(), the headline line as the driver runs it, and the kernel trace of the same loop
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03_v15_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_v15_pytest.log
grep -n "passed\|failed\|error" gpurun_out/r03_v15_pytest.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_v15_bench.json 2> gpurun_out/r03_v15_bench.err; echo "bench rc=$?"
python -c "
import json;d=json.loads(open('gpurun_out/r03_v15_bench.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d.get('sustained_200_steps'), d['roofline'], d['cpu_baseline'])"
rm -rf gpurun_out/prof_r03_v15
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v15 -o bench -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v15_bench_prof.log 2>&1
DB=$(ls gpurun_out/prof_r03_v15/*.db gpurun_out/prof_r03_v15/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/r03_v15_bench_kernels_bf16.txt 2>&1; head -12 gpurun_out/r03_v15_bench_kernels_bf16.txt; fi
rm -rf gpurun_out/prof_r03_v15
