#!/bin/bash
# rocprofv3 counter passes over a few fused-step launches (counters only with --kernel-trace: see gpurun rules)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc/p$i -o pmc -f csv -- python tools/run_step_once.py ${1:-8192} ${2:-f32} > gpurun_out/pmc/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in agg.items():
            if "rlx" in k:
                print(d, k, {c: round(x / 5) for c, x in v.items()})
PY
