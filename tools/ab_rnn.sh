for lib in "$@"; do
  NNV_LIB=$PWD/neural-network-based-viterbi-decoding_amd/$lib python tools/bench_rnn.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'f32 ms', round(d['ms_per_step'],3), 'f16x3 ms', round(d['fast_mode']['ms_per_step'],3), 'err', d['parity_sample']['max_abs_logit_error_vs_float64'])"
done
