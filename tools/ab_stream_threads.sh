echo "--- default"; tools/ab_configs01.sh libnnviterbi_hip.so
echo "--- 768 threads, 16 KB slots, paired chains"; NNV_THREADS=768 tools/ab_configs01.sh ab_p768r16.so
echo "--- 768 threads, 8 KB slots"; NNV_THREADS=768 tools/ab_configs01.sh ab_p768r8.so
echo "--- 1024 threads, 16 KB slots, paired chains"; NNV_THREADS=1024 tools/ab_configs01.sh ab_p1024r16.so
echo "--- default again"; tools/ab_configs01.sh libnnviterbi_hip.so
