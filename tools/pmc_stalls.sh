#!/bin/bash
# Why do the mixed read/write streams stop at 0.62-0.67 of the HBM peak?  One rocprofv3 --pmc pass per counter (kernel-trace only:
# the gpurun rules), counters that this rocprofv3 does not know are skipped.  Output: gpurun_out/pmc_stalls.txt
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/rocprofv3_counters_available.txt 2>&1
for C in SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum MeanOccupancyPerCU; do
  if ! grep -q "$C" gpurun_out/rocprofv3_counters_available.txt; then echo "skip $C (not listed)"; continue; fi
  rm -rf gpurun_out/pmc_s_$C
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmc_s_$C -o pmc -f csv -- python tools/pmc_stalls_probe.py > gpurun_out/pmc_s_$C.log 2>&1 || echo "pass $C failed"
done
python - <<'PY' | tee gpurun_out/pmc_stalls.txt
import csv, glob, collections, os
res = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc_s_*/")):
    C = os.path.basename(os.path.dirname(d))[len("pmc_s_"):]
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != C:
                continue
            name = row["Kernel_Name"]
            for key in ("copyBuffer", "bfloat16_copy", "elementwise", "token_logprob_fwd", "token_logprob_bwd", "gae_seq", "gae_scan"):
                if key in name:
                    acc[key].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[k][C] = sum(v) / len(v)
cols = sorted({c for v in res.values() for c in v})
print("per-launch means; kernels:", ", ".join(res))
for c in cols:
    print(f"{c:36s} " + "  ".join(f"{k}={res[k].get(c, float('nan')):.4g}" for k in res))
PY
rm -rf gpurun_out/pmc_s_*/
