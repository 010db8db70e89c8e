echo "--- default threads"; tools/ab_configs01.sh libnnviterbi_hip.so ab_ring16.so
echo "--- NNV_THREADS=1024"; NNV_THREADS=1024 tools/ab_configs01.sh libnnviterbi_hip.so ab_ring16.so
echo "--- NNV_THREADS=512"; NNV_THREADS=512 tools/ab_configs01.sh libnnviterbi_hip.so ab_ring16.so
