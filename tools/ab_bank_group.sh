#!/bin/bash
# NOTE (round 5): NNV_BANK_GROUP is read only by libraries built with tools/build_variant.sh (-DNNV_AB_BUILD);
# point NNV_LIB at such a build -- the shipped libnnviterbi_hip.so reads no environment variable.
# Bit-wise bank tile order A/B on the GPU box: NNV_BANK_GROUP = column blocks per L2-resident W slab (0 = round-2 order)
for g in ${@:-0 4 6 7 8 12}; do
  NNV_BANK_GROUP=$g python tools/bench_bitwise.py --steps 8 > /tmp/bank.json 2>/tmp/bank.err || tail -3 /tmp/bank.err
  python - $g <<'PY'
import json,sys
b=json.load(open("/tmp/bank.json"))["bank"]
print("group", sys.argv[1], "ms", round(b["ms_per_step"],3), "frac", round(b["roofline"]["frac"],4), "mismatches", b["parity_sample"]["class_mismatches"])
PY
done
