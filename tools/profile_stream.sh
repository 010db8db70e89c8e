#!/bin/bash
# Runs ON the GPU box: evidence for the K1 STREAM variant (BASELINE configs[0]/[1], tools/bench_configs01.py):
#   tools/profile_stream.sh <tag>   -> gpurun_out/prof_<tag>/{trace_c01, pmc3_*}
# kernel-trace stats first, then separate --pmc passes (never combined with tracing).  Summarise with
#   tools/summarize_pmc.py --stream gpurun_out/prof_<tag> profiles/<round>/pmc_stream.json
set -u
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c01 -o run -- python3 tools/bench_configs01.py --steps 20 > $out/bench_configs01_under_rocprof.json 2> $out/trace_c01.log
echo "trace done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-32)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc3_$name -o run -- python3 tools/bench_configs01.py --steps 3 > /dev/null 2> $out/pmc3_$name.log
  echo "pmc3 $grp done"
done
find $out -name "*kernel_trace.csv" -size +3M -delete
