#!/bin/bash
# Usage: tools/shape_sweep.sh lib1.so lib2.so ...  -- K1 over the tile sizes b = 6 .. 10 (8 M rows each), one line per shape
for lib in "$@"; do
for d in "14,128,64,32,128" "16,128,64,32,256" "18,128,64,32,512" "12,128,64,32,64" "20,128,64,1024"; do NNV_LIB=$PWD/neural-network-based-viterbi-decoding_amd/$lib python tools/bench_band.py $d 8000000 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['net'], {k:(round(d[k]['ms_per_launch'],3), round(d[k]['tflops_algorithmic'],1)) for k in ('f32','f16x3')})"; done; done
