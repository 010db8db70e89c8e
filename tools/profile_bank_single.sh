#!/bin/bash
# Runs ON the GPU box: counters of ONE bit-wise position, K4d (bank_single_kernel) and K4b (dense_tanh_kernel<..32-column
# tile, HEAD>) side by side inside tools/bench_bank_small.py -- separate --pmc passes, never combined with tracing --
# and the kernel-trace statistics of the same program.
#   tools/profile_bank_single.sh <tag>  -> gpurun_out/prof_<tag>/pmc_single_*/ ; summary on stdout (JSON)
set -u
tag=${1:-r04}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_single -o run -- python3 tools/bench_bank_small.py --positions 1 --steps 20 > /dev/null 2> $out/stats_single.log
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-32)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_single_$name -o run -- python3 tools/bench_bank_small.py --positions 1 --steps 4 > /dev/null 2> $out/pmc_single_$name.log
done
python3 - $out <<'PY'
import csv, glob, json, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/pmc_single_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        n = r["Kernel_Name"]
        key = "K4d bank_single_kernel<9>" if "bank_single_kernel" in n else ("K4b dense_tanh_kernel, 32-column tile + head" if "dense_tanh_kernel" in n else None)
        if key:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    d = {"per_launch_means": m, "launches": {n: len(v) for n, v in c.items()}}
    if "FETCH_SIZE" in m:
        d["l2_from_fabric_GB_per_launch_x2_corrected"] = m["FETCH_SIZE"] * 2048 / 1e9
    if "WRITE_SIZE" in m:
        d["written_GB_per_launch"] = m["WRITE_SIZE"] * 1024 / 1e9
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        d["mfma_busy_frac_per_simd"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)
    res[k] = d
stats = {}
for path in glob.glob(sys.argv[1] + "/stats_single/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        if "bank_single_kernel" in r["Name"] or "dense_tanh_kernel" in r["Name"]:
            stats[r["Name"][:90]] = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"])}
res["kernel_trace_stats"] = stats
res["algorithmic_bytes_per_launch"] = 300000 * (4 * 1036 + 1)
print(json.dumps(res, indent=1))
PY
