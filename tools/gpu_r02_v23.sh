#!/bin/bash
# Round 2, GPU visit 23: gae_seq per-sequence walk with smaller LDS segments (threads x segment sweep) on the three shapes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 10 11 12 13 14 15 16; do
RLX_GAESEQ_VARIANT=$v timeout 300 python tools/bench_widening.py > gpurun_out/v23_wid_$v.log 2>&1
echo "variant=$v rc=$? $(grep '"kernel": "gae_seq' gpurun_out/v23_wid_$v.log | python -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["shape"], d["frac"], end="  ")')"
done
RLX_GAESEQ_VARIANT=11 timeout 600 python -m pytest tests/test_gpu_token_path.py -q -m gpu -x -k "gae_seq" > gpurun_out/v23_t.log 2>&1; echo "t rc=$?"; tail -1 gpurun_out/v23_t.log
