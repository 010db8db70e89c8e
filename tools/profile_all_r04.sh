set -u
# Round 4's evidence in one gpurun call: headline trace + PMC, stream variant, K6 (sliding window), secondary paths.
tools/profile_round.sh r04 > gpurun_out/prof_r04.log 2>&1; tail -3 gpurun_out/prof_r04.log
tools/profile_stream.sh r04 > gpurun_out/prof_r04_stream.log 2>&1; tail -2 gpurun_out/prof_r04_stream.log
tools/profile_fused.sh r04 > gpurun_out/prof_r04_fused.log 2>&1; tail -2 gpurun_out/prof_r04_fused.log
