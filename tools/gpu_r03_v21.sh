#!/bin/bash
# round 3, visit 21: re-check two tuning knobs of the update phase after this round's LDS-layout and loop changes (same box, 40 iterations each)
set -u
mkdir -p gpurun_out
run() { # name=value ...
  env "$@" timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$*', d['ms_per_step'], d['ms_per_step_windows'])" | tee -a gpurun_out/r03_v21_knob_sweep.txt
}
run RLX_DW_NBUF=3
run RLX_DW_NBUF=2
run RLX_DW_NBUF=4
run RLX_DW_NBUF=5
run RLX_DW_NBUF=6
run RLX_FUSED_PD=2
run RLX_FUSED_PD=3
run RLX_FUSED_PD=4
run RLX_DW_NBUF=3
