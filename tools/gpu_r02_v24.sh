#!/bin/bash
# Round 2, GPU visit 24: reinpp returns kernel lanes-per-sequence sweep; gae_seq short-row variants.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -x -k "reinpp" > gpurun_out/v24_t.log 2>&1; echo "t_reinpp rc=$?"; tail -1 gpurun_out/v24_t.log
for rt in 256 128 64 512 256; do
RLX_REINPP_RT=$rt timeout 300 python tools/bench_widening.py > gpurun_out/v24_wid_rt$rt.log 2>&1
echo "reinpp_rt=$rt rc=$? $(grep '"kernel": "reinpp' gpurun_out/v24_wid_rt$rt.log | python -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["shape"], d["frac"], end="  ")')"
done
for v in 0 20 21 22; do
RLX_GAESEQ_VARIANT=$v timeout 300 python tools/bench_widening.py > gpurun_out/v24_wid_g$v.log 2>&1
echo "gaeseq_variant=$v rc=$? $(grep '"kernel": "gae_seq' gpurun_out/v24_wid_g$v.log | python -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["shape"], d["frac"], end="  ")')"
done
for rt in 128 64; do RLX_REINPP_RT=$rt timeout 600 python -m pytest tests -q -m gpu -x -k "reinpp" > gpurun_out/v24_t_rt$rt.log 2>&1; echo "t_reinpp_rt$rt rc=$?"; tail -1 gpurun_out/v24_t_rt$rt.log; done
