#!/bin/bash
# Usage: tools/ab_bank_small.sh lib1.so lib2.so ...  -- small bit-wise banks (tools/bench_bank_small.py) with each library
for lib in "$@"; do
  NNV_LIB="$PWD/neural-network-based-viterbi-decoding_amd/$lib" python tools/bench_bank_small.py --positions 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for l in d['legs']: print('$lib', 'P=%d default %.4f ms (%.3f of HBM peak) | K4b %.4f ms' % (l['positions'], l['default_ms'], l['default_hbm_frac_per_position_pass'], l['tiled_K4b_ms']))"
done
