# same-box A/B of two builds of libsan_hip.so: usage  bash scratch/ab_lib.sh <script and args...>
# (scratch/libsan_old.so = baseline build, in-tree libsan_hip.so = candidate); old | new | new (repeat)
cp spatialalignmentnetwork_amd/libsan_hip.so /tmp/new.so
python "$@" > /tmp/new.txt 2>/dev/null
cp scratch/libsan_old.so spatialalignmentnetwork_amd/libsan_hip.so
python "$@" > /tmp/old.txt 2>/dev/null
cp /tmp/new.so spatialalignmentnetwork_amd/libsan_hip.so
python "$@" > /tmp/new2.txt 2>/dev/null
paste -d"|" <(cut -c1-46 /tmp/old.txt) <(cut -c22-46 /tmp/new.txt) <(cut -c22-46 /tmp/new2.txt)
