#!/bin/bash
# rocprofv3 counter passes (own runs, --kernel-trace only: the gpurun rules) over a few launches of the fused optimizer step and the
# rollout step: what the L2 -> CU path of the fused launch does.  -> profiles/r06_fused_l2_counters.txt
#   bash tools/pmc_fused_l2.sh [rows=8192] [bf16|f32]
set -u
export TMPDIR=/tmp
out=gpurun_out/pmc_l2
mkdir -p $out
i=0
# (at most four counters of one hardware block per pass: a bigger request is refused -- "exceeds the capabilities of the hardware" --
# and the refused run then sits until the timeout)
for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "TCC_READ_sum TCC_WRITE_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" \
           "TCC_BUSY_sum TCC_TAG_STALL_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rm -rf $out/p$i
  timeout 100 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o pmc -f csv -- python tools/run_step_once.py ${1:-8192} ${2:-bf16} > $out/p$i.log 2>&1
  echo "pass $i rc=$? ($set)"
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_l2/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
        for k, v in agg.items():
            if "rlx" in k:
                print(d, k, "launches", len(n[k]), {c: round(x / max(1, len(n[k]))) for c, x in v.items()})
PY
