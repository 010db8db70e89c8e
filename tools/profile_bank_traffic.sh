#!/bin/bash
# NOTE (round 5): NNV_BANK_GROUP is read only by libraries built with tools/build_variant.sh (-DNNV_AB_BUILD);
# point NNV_LIB at such a build -- the shipped libnnviterbi_hip.so reads no environment variable.
# Runs ON the GPU box: L2 <- fabric bytes (FETCH_SIZE) of the bit-wise bank kernel per tile order.
#   tools/profile_bank_traffic.sh <tag> [groups...]  -> gpurun_out/prof_<tag>/pmc_bank_g<G>/
set -u
tag=${1:-r03}; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
for g in ${@:-0 7}; do
  NNV_BANK_GROUP=$g rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_bank_g$g -o run -- python3 tools/bench_bitwise.py --steps 4 > /dev/null 2> $out/pmc_bank_g$g.log
  python3 - $out/pmc_bank_g$g $g <<'PY'
import csv, glob, sys
vals=[]
for path in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        if "dense_tanh_kernel<true, 32, 2, 128, true>" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
            vals.append(float(r["Counter_Value"]))
print(f"group {sys.argv[2]}: {len(vals)} launches, FETCH_SIZE mean {sum(vals)/max(len(vals),1):.0f} KB -> x2 corrected {sum(vals)/max(len(vals),1)*2048/1e9:.2f} GB per launch")
PY
done
