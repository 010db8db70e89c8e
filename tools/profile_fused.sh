#!/bin/bash
# Runs ON the GPU box: kernel-trace stats, then separate --pmc passes (never combined with tracing), of the
# sliding-window benchmark (K6 wide_fused_kernel by default, K4 beside it inside the same program):
#   tools/profile_fused.sh <tag>   -> gpurun_out/prof_<tag>/{stats_seq, pmc2_seq_predict_*}
# Summarise with  tools/summarize_pmc.py --secondary gpurun_out/prof_<tag> profiles/<round>/pmc_fused.json
set -u
tag=${1:-r04}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_seq -o run -- python3 tools/bench_seq_predict.py --steps 10 > $out/bench_seq_under_rocprof.json 2> $out/stats_seq.log
echo "stats done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-32)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc2_seq_predict_$name -o run -- python3 tools/bench_seq_predict.py --steps 2 > /dev/null 2> $out/pmc2_seq_predict_$name.log
  echo "pmc2 seq_predict $grp done"
done
