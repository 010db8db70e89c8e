#!/bin/bash
# Usage: [NNV_PREC=f32|f16x3] tools/ab_bench.sh lib1.so lib2.so ...   (paths relative to the package dir)
# Runs bench.py (no CPU baseline) with each library and prints ms/step + roofline fraction.
for lib in "$@"; do
  NNV_LIB="$PWD/neural-network-based-viterbi-decoding_amd/$lib" NNV_BENCH_SMALL_LEGS=1 python bench.py --no-cpu-baseline --no-live-traffic --steps 10 --warmup 2 --precision ${NNV_PREC:-f32} > /tmp/ab.json 2>/tmp/ab.err || tail -5 /tmp/ab.err
  python - "$lib" <<'PY'
import json,sys
d=json.load(open("/tmp/ab.json"))
print(f"{sys.argv[1]:24s} {d['dtype']:6s} ms/step {d['ms_per_step']:.3f}  kernel {d['roofline']['kernel_ms_per_launch']:.3f}  Mblocks/s {d['value']/1e6:.1f}  frac {d['roofline']['frac']:.4f}  threads {d['launch']['threads']} vgprs {d['launch']['vgprs']}")
PY
done
