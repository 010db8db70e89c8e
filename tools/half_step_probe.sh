for lib in libnnviterbi_hip.so ab_half.so libnnviterbi_hip.so ab_half.so; do
NNV_LIB=$PWD/neural-network-based-viterbi-decoding_amd/$lib python tools/bench_wide_shapes.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if d['net'][1]==512 and d['net'][2]==512 or d['net'][0]==200: print('$lib', d['net'], 'K6 %.3f ms %.3f' % (d['fused_K6']['ms'], d['fused_K6']['frac_of_fp32_mfma_peak']))"
done
