#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one --pmc pass each, kernel-trace only) of the widening kernels, next to a copy of
# known size for the unit calibration.  Output: gpurun_out/pmc_widening.txt
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_w_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmc_w_$C -o pmc -f csv -- python tools/pmc_widening_probe.py > gpurun_out/pmc_w_$C.log 2>&1
done
python - <<'PY' | tee gpurun_out/pmc_widening.txt
import csv, glob, collections
res = collections.defaultdict(dict)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_w_{C}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != C:
                continue
            name = row["Kernel_Name"]
            for key in ("copyBuffer", "bfloat16_copy", "token_logprob_fwd", "token_logprob_bwd", "patch_scan", "gae_seq",
                        "reinpp_returns", "reinpp_normalize", "copy_segments"):
                if key in name:
                    acc[key].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[k][C] = sum(v) / len(v)
NB = 2048 * 151936 * 2
cal = next((res[k] for k in ("copyBuffer", "bfloat16_copy") if k in res), None)
print("kernel raw_FETCH raw_WRITE  read_MB write_MB  (calibrated on the copy: both directions =", NB / 1e6, "MB)")
for k, v in res.items():
    f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    rd = f * NB / cal["FETCH_SIZE"] / 1e6 if cal and cal.get("FETCH_SIZE") else float("nan")
    wr = w * NB / cal["WRITE_SIZE"] / 1e6 if cal and cal.get("WRITE_SIZE") else float("nan")
    print(f"{k:20s} {f:12.1f} {w:12.1f} {rd:10.1f} {wr:10.1f}")
print("algorithmic MB: fwd read", NB / 1e6, "; bwd read+write", NB / 1e6, "each; patch_scan read", 2 * NB / 1e6,
      "; gae_seq read", 4096 * 8192 * 4 / 1e6, "write", 4096 * 8192 * 8 / 1e6,
      "; reinpp_returns read", 4096 * 8192 * 10 / 1e6, "write", 4096 * 8192 * 4 / 1e6, "; reinpp_normalize read = write",
      4096 * 8192 * 4 / 1e6, "; copy_segments read", 8 * 4096 * 8192 * 4 / 1e6, "write", 8 * 4096 * 8192 * 2 / 1e6)
PY
