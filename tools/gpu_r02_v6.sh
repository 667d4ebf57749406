#!/bin/bash
# Round 2, GPU visit 6: gae_seq with 1024-token look-back segments, reinpp wave-per-row kernel, pipeline benches after the numpy perms.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -4 gpurun_out/$name.log | cut -c1-300; }
run v6_t_token 600 python -m pytest tests/test_gpu_token_path.py -q -m gpu -k "gae_seq or reinpp or empty_and_degenerate"
run v6_t_pipe 600 python -m pytest tests/test_end_to_end.py -q -m gpu -k "pipeline"
run v6_widening 300 python tools/bench_widening.py; grep kernel gpurun_out/v6_widening.log
rm -f gpurun_out/v6_pipeline_bench.log
for args in "" "--pipeline" "--pipeline --rollout-epochs 2" "--pipeline --rollout-epochs 2 --no-overlap" "--pipeline --rollout-epochs 4" "--pipeline --rollout-epochs 4 --no-overlap"; do
  timeout 300 python bench.py $args --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$args]', d['ms_per_step'], d['value'])" | tee -a gpurun_out/v6_pipeline_bench.log
done
