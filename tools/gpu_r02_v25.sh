#!/bin/bash
# Round 2, GPU visit 25: verification of the final state -- full GPU suite, smoke, bench (full line), widening sweep.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout "$1" "${@:2}" > gpurun_out/$name.log 2>&1; echo "$name rc=$?"; tail -2 gpurun_out/$name.log | cut -c1-220; }
run v25_t_all 1800 python -m pytest tests -q -m gpu -x
run v25_smoke 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
run v25_bench_a 900 python bench.py
run v25_widening 300 python tools/bench_widening.py
grep '"kernel"' gpurun_out/v25_widening.log | cut -c1-200
