#!/bin/bash
# Runs ON the GPU box: every benchmark record of the round -> gpurun_out/<tag>/*.json
#   tools/run_round_benches.sh r02 [quick]
# Copy what is to be judged into profiles/<tag>/ afterwards.  Each leg is one python process.
set -u
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
run() {  # stdout -> <name>.json (JSON lines only: gloo prints banners on stdout), stderr -> <name>.log
  name=$1; shift; echo "== $name: $*"
  "$@" > $out/$name.raw 2> $out/$name.log || echo "   FAILED rc=$? (see $out/$name.log)"
  grep '^{' $out/$name.raw > $out/$name.json; rm -f $out/$name.raw
  tail -c 400 $out/$name.json; echo
}
run bench_n1_default python3 bench.py
NNV_DIST_BACKEND=gloo run bench_2rank_rehearsal_one_gpu python3 bench.py --gpus 2 --blocks-per-gpu 4000000 --steps 5 --warmup 2 --no-cpu-baseline
run bench_bitwise python3 tools/bench_bitwise.py
run host_path python3 tools/host_path_bench.py
echo '== flip_rates'; python3 tools/flip_rates.py > $out/flip_rates.json 2> $out/flip_rates.log || echo '   FAILED'   # indented JSON: not line-filtered
run bench_config4_full_share python3 tools/bench_config4.py --blocks-per-gpu 12500000
run bench_configs01 python3 tools/bench_configs01.py
run bench_configs01_random_weights python3 tools/bench_configs01.py --weights random
run bench_seq_predict python3 tools/bench_seq_predict.py
run bench_rnn_10M_blocks python3 tools/bench_rnn.py --blocks 10000000 --steps 2
run bench_batch_sweep python3 tools/bench_batch_sweep.py
