#!/bin/bash
# Runs ON the GPU box (via gpurun): collects the evidence bench.py's roofline object cites.
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/...
# 1. rocprofv3 --kernel-trace --stats of the default `python3 bench.py` (the command the driver runs)
# 2. separate --pmc passes (never combined with tracing): FETCH_SIZE | WRITE_SIZE | SQ groups
# 3. the secondary paths' kernel stats (sliding-window, recurrent, bit-wise bank)
# Summarise afterwards with tools/summarize_pmc.py and copy what is to be judged into profiles/.
set -u
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py"   # the driver's own default command (K = 20, W = 3, cpu_baseline leg included)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- $BENCH > $out/bench_under_rocprof.json 2> $out/trace.log
echo "trace done: $(tail -c 300 $out/bench_under_rocprof.json | head -c 200)"
# counter passes: the headline kernel is what is summarised; the other legs run at their reduced sizes
# (bench.py itself starts no live counter passes when it sees a profiler's environment)
export NNV_BENCH_SMALL_LEGS=1
PMCBENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$name -o run -- $PMCBENCH > /dev/null 2> $out/pmc_$name.log
  echo "pmc $grp done"
done
unset NNV_BENCH_SMALL_LEGS
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_seq -o run -- python3 tools/bench_seq_predict.py > $out/bench_seq_predict.json 2> $out/trace_seq.log
echo "seq done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_rnn -o run -- python3 tools/bench_rnn.py > $out/bench_rnn.json 2> $out/trace_rnn.log
echo "rnn done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_bitwise -o run -- python3 tools/bench_bitwise.py > $out/bench_bitwise.json 2> $out/trace_bitwise.log
echo "bitwise done"
find $out -name "*kernel_trace.csv" -size +3M -delete   # keep the summaries, not multi-MB traces
