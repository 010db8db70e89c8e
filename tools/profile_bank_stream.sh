#!/bin/bash
# Runs ON the GPU box: counters of the fp32 bit-wise bank, K4c (bank_stream_kernel) and K4b (dense_tanh_kernel<..HEAD>)
# side by side inside tools/bench_bitwise.py -- separate --pmc passes, never combined with tracing.
#   tools/profile_bank_stream.sh <tag>  -> gpurun_out/prof_<tag>/pmc_bank_*/ ; summary on stdout (JSON)
set -u
tag=${1:-r04}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
for grp in "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-32)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_bank_$name -o run -- python3 tools/bench_bitwise.py --steps 4 > /dev/null 2> $out/pmc_bank_$name.log
done
python3 - $out <<'PY'
import csv, glob, json, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/pmc_bank_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        n = r["Kernel_Name"]
        key = "K4c bank_stream_kernel" if "bank_stream_kernel" in n else ("K4b dense_tanh_kernel<true,32,2,128,true>" if "dense_tanh_kernel<true, 32, 2, 128, true>" in n else None)
        if key:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    d = {"per_launch_means": m, "launches": {n: len(v) for n, v in c.items()}}
    if "FETCH_SIZE" in m:
        d["l2_from_fabric_GB_per_launch_x2_corrected"] = m["FETCH_SIZE"] * 2048 / 1e9
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        d["mfma_busy_frac_per_simd"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)
    res[k] = d
print(json.dumps(res, indent=1))
PY
