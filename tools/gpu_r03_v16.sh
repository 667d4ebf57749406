#!/bin/bash
# round 3, visit 16: the run-ahead loop in pipeline mode (perm uploads from pinned staging buffers)
set -u
mkdir -p gpurun_out
for e in 4 1 0; do
echo "pipeline epochs $e"
timeout 600 python tools/ab_defer.py --steps 30 --rounds 2 --pipeline-epochs $e 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_v16_ab_defer_pipeline.txt
done
timeout 900 python -m pytest tests/test_end_to_end.py -m gpu -q -p no:cacheprovider -k "pipeline or run_ahead" 2>&1 | tail -3
