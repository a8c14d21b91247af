# ablation diagnostic of the bf16x3 convolution (scratch/libs/abl.so, built from a copy of the kernel with run-time ablation hooks):
#   SAN_B16_ABL bit 1 = no activation loads, 2 = no weight loads, 4 = no K-loop, 8 = no statistics, 16 = no stores
cp spatialalignmentnetwork_amd/libsan_hip.so /tmp/keep.so
cp scratch/libs/abl.so spatialalignmentnetwork_amd/libsan_hip.so
for m in ${ABLS:-0 31 63 127 95 255 223 160 64 32}; do
  echo "== ABL=$m"; SAN_B16_ABL=$m BL_ONLY=${1:-18-18-320,36-18-320,72-36-160,144-72-80} python scratch/bench_layers.py conv 2>/dev/null | grep "@"
done
cp /tmp/keep.so spatialalignmentnetwork_amd/libsan_hip.so
