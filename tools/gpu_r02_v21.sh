#!/bin/bash
# Round 2, GPU visit 21: gae_scan streaming variant with XCD-paired env groups (done lines fetched once).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_advantages.py -q -m gpu -x > gpurun_out/v21_t_adv.log 2>&1; echo "t_adv rc=$?"; tail -1 gpurun_out/v21_t_adv.log
for rep in 1 2; do
for pair in 0 1; do
RLX_GAE_PAIR=$pair timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/v21_bench_pair$pair.log 2>&1
echo "pair=$pair rc=$? $(tail -1 gpurun_out/v21_bench_pair$pair.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(r["frac"], r["avg_launch_us"], r["traffic"], d["ms_per_step"])' 2>&1 | tail -1)"
done
done
