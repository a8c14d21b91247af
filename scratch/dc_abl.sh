# ablation of dc_rows_kernel at the 15-coil 640 x 368 shape (variant library scratch/libs/fftabl.so: SAN_DC_ABL bits
# 1 no forward transform, 2 no inverse, 4 no k0 load, 8 no epilogue loads, 16 no store, 32 no radix-23 pass; SAN_DC_B rows/WG)
cp scratch/libs/fftabl.so spatialalignmentnetwork_amd/libsan_hip.so
for abl in 0 1 2 3 32 35 4 8 16 28 31 63; do
  echo "ABL=$abl: $(SAN_DC_ABL=$abl python scratch/bench_dc_rows.py 1 15 640 368 2>&1 | tail -1)"
done
for b in 1 2 4 8 16; do
  echo "B=$b: $(SAN_DC_B=$b python scratch/bench_dc_rows.py 1 15 640 368 2>&1 | tail -1)"
done
