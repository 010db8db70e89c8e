#!/bin/bash
# Runs ON the GPU box: PMC passes (never combined with tracing) for the secondary paths,
#   tools/profile_secondary.sh <tag>   -> gpurun_out/prof_<tag>/pmc2_*/
# K4 (tools/bench_seq_predict.py), K5 (tools/bench_rnn.py) and the bit-wise bank (tools/bench_bitwise.py).  Summarise with
#   tools/summarize_pmc.py --secondary gpurun_out/prof_<tag> profiles/<round>/pmc_secondary.json
set -u
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
for prog in "tools/bench_seq_predict.py --steps 2" "tools/bench_rnn.py --steps 1" "tools/bench_bitwise.py --steps 4"; do
  short=$(echo $prog | sed 's#tools/bench_##; s#\.py.*##')
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS"; do
    name=$(echo $grp | tr ' ' '_' | cut -c1-32)
    rocprofv3 --pmc $grp --output-format csv -d $out/pmc2_${short}_$name -o run -- python3 $prog > /dev/null 2> $out/pmc2_${short}_$name.log
    echo "pmc2 $short $grp done"
  done
done
