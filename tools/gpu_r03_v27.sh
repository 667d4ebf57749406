#!/bin/bash
# round 3, visit 27: the driver-style line once more on whatever box this visit draws (box-to-box spread of the headline)
set -u
mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_v27_bench.json 2> gpurun_out/r03_v27_bench.err; echo "bench rc=$?"
python -c "
import json;d=json.loads(open('gpurun_out/r03_v27_bench.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['sustained_200_steps']['ms_per_step'], d['sustained_200_steps']['ms_per_step_windows'], d['roofline']['frac'], [v['ms_per_step'] for v in d['variants']], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
