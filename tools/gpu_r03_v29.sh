#!/bin/bash
# round 3, visit 29: soak -- the run-ahead loop over thousands of iterations (pinned-buffer rings, events, graph replays), sync and pipeline mode
set -u
mkdir -p gpurun_out
timeout 600 python bench.py --gpus 1 --steps 3000 --warmup 5 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v29_soak_sync.json 2>/dev/null; echo "sync rc=$?"
timeout 600 python bench.py --gpus 1 --steps 1500 --warmup 5 --pipeline --rollout-epochs 4 --no-cpu-baseline --no-roofline --no-variants > gpurun_out/r03_v29_soak_pipeline_e4.json 2>/dev/null; echo "pipeline rc=$?"
python - <<'PY'
import json
for f in ("sync", "pipeline_e4"):
    d = json.loads(open(f"gpurun_out/r03_v29_soak_{f}.json").read().strip().splitlines()[-1])
    print(f, d["steps"], d["ms_per_step"], d["ms_per_step_windows"], d["last_metrics"])
PY
