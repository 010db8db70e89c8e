set -u
tools/profile_round.sh r03 > gpurun_out/prof_r03.log 2>&1; tail -3 gpurun_out/prof_r03.log
tools/profile_stream.sh r03 > gpurun_out/prof_r03_stream.log 2>&1; tail -2 gpurun_out/prof_r03_stream.log
tools/profile_secondary.sh r03 > gpurun_out/prof_r03_secondary.log 2>&1; tail -2 gpurun_out/prof_r03_secondary.log
