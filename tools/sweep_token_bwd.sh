#!/bin/bash
# Dev sweep: token_logprob_bwd variants (unroll, non-temporal loads / stores), rebuilt on the GPU box.
set -u
mkdir -p gpurun_out
for v in "-DRLX_TOK_UNROLL=4" "-DRLX_TOK_UNROLL=2" "-DRLX_TOK_UNROLL=8" "-DRLX_TOK_BWD_NT_STORE=0" "-DRLX_TOK_BWD_NT_LOAD=0" "-DRLX_TOK_BWD_NT_LOAD=0 -DRLX_TOK_BWD_NT_STORE=0"; do
  touch rlinf_amd/csrc/token_ops.hip
  RLX_CXXFLAGS="$v" python -m rlinf_amd.csrc.build > /dev/null 2>&1
  echo "== $v"
  python tools/bench_token.py --tokens 8192 --vocab 151936 --dtype bf16 --iters 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if not d['entropy']: print(d['kernel'], round(d['us'],1), round(d['frac'],3))"
done | tee gpurun_out/token_bwd_sweep.txt
