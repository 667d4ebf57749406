#!/bin/bash
# Round 2, GPU visit 5: token-path kernels after the gae_seq look-back / reinpp pipelining, the multi-epoch pipeline learner,
# the real-registry hook, bench --pipeline variants.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -4 gpurun_out/$name.log | cut -c1-300; }
run v5_t_token 600 python -m pytest tests/test_gpu_token_path.py -q -m gpu -k "gae_seq or reinpp or empty_and_degenerate"
run v5_t_pipe 600 python -m pytest tests/test_end_to_end.py tests/test_gpu_ext_real_registry.py -q -m gpu
run v5_widening 300 python tools/bench_widening.py; cat gpurun_out/v5_widening.log | grep kernel
RLX_GAESEQ_VARIANT=1 timeout 300 python tools/bench_widening.py 2>/dev/null | grep gae_seq | tee gpurun_out/v5_widening_old.log
for args in "--pipeline" "--pipeline --rollout-epochs 2" "--pipeline --rollout-epochs 2 --no-overlap" "--pipeline --rollout-epochs 4" "--pipeline --rollout-epochs 4 --no-overlap"; do
  timeout 300 python bench.py $args --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args', d['ms_per_step'], d['value'])" | tee -a gpurun_out/v5_pipeline_bench.log
done
