#!/bin/bash
# round 3, visit 23 (final): the full GPU suite, smoke(), the headline line as the driver runs it, and the kernel trace of the same
# command with the roofline probe in it (so that the graded kernel's launches are in the trace)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03_v23_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_v23_pytest.log
grep -n "passed\|failed\|error" gpurun_out/r03_v23_pytest.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_v23_bench.json 2> gpurun_out/r03_v23_bench.err; echo "bench rc=$?"
python -c "
import json;d=json.loads(open('gpurun_out/r03_v23_bench.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d.get('sustained_200_steps'), d['roofline']['frac'], d['roofline']['avg_launch_us'], [ (v['ms_per_step']) for v in d['variants']])"
rm -rf gpurun_out/prof_r03_v23
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_v23 -o bench -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-token-tier --no-traffic > gpurun_out/r03_v23_bench_prof.log 2>&1
DB=$(ls gpurun_out/prof_r03_v23/*.db gpurun_out/prof_r03_v23/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py "$DB" > gpurun_out/r03_v23_bench_kernels_bf16.txt 2>&1; head -8 gpurun_out/r03_v23_bench_kernels_bf16.txt | cut -c1-150; grep "gae_scan" gpurun_out/r03_v23_bench_kernels_bf16.txt | cut -c1-150; fi
rm -rf gpurun_out/prof_r03_v23
