echo "--- default"; tools/ab_configs01.sh libnnviterbi_hip.so
echo "--- NNV_THREADS=512 ring32"; NNV_THREADS=512 tools/ab_configs01.sh ab_ring32.so
echo "--- NNV_THREADS=768 ring32 (f16 only has 768)"; NNV_THREADS=768 tools/ab_configs01.sh ab_ring32.so
echo "--- default again"; tools/ab_configs01.sh libnnviterbi_hip.so
