# same-box comparison of several builds of libsan_hip.so kept under scratch/libs/:  bash scratch/abc_libs.sh "old ring2 ring3" <script and args>
LIBS=$1; shift
cp spatialalignmentnetwork_amd/libsan_hip.so /tmp/keep.so
first=1
for l in $LIBS; do
  cp scratch/libs/$l.so spatialalignmentnetwork_amd/libsan_hip.so
  cp include/san_hip.h /tmp/keep.h
  if [ -f scratch/libs/$l.hide ]; then for sym in $(cat scratch/libs/$l.hide); do sed -i "/$sym/d" include/san_hip.h; done; fi   # entry points this build predates
  python "$@" > /tmp/abc_$l.txt 2>/dev/null
  cp /tmp/keep.h include/san_hip.h
done
cp /tmp/keep.so spatialalignmentnetwork_amd/libsan_hip.so
set -- $LIBS
cmd="paste -d'|' <(cut -c1-32 /tmp/abc_$1.txt)"; shift
for l in "$@"; do cmd="$cmd <(cut -c22-32 /tmp/abc_$l.txt)"; done
eval $cmd
