#!/bin/bash
# K4c (rows resident, first layers streamed) against K4b (tiled GEMM) over the number of positions in the bank
for P in "$@"; do
  python tools/bench_bitwise.py --steps 8 --positions $P 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['bank']
print('P=%4d  default path %.3f ms | K4b %.3f ms | same bits %s | single_fused %.3f ms' % ($P, b['ms_per_step'], b['tiled_gemm_K4b']['ms_per_step'], b['tiled_gemm_K4b']['decided_bits_identical'], d['single_fused']['ms_per_step']))"
done
