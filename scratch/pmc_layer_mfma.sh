# Hardware counters of ONE layer's convolution launches (scratch/bench_layers.py, BL_ONLY=cin-cout-size), three --pmc passes:
#   bash scratch/pmc_layer.sh 36-36-160 [conv|wgrad]   ->  gpurun_out/pmc_layer_<layer>.txt (per-launch averages)
L=${1:-36-36-160}; W=${2:-conv}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_LEVEL_VMEM"
P4="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1)); rm -rf /tmp/pl$i
  BL_ONLY=$L timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pl$i -o p -- python $R/scratch/bench_layers.py $W > /tmp/pl$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in ('/tmp/pl1', '/tmp/pl2', '/tmp/pl3', '/tmp/pl4'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if "conv_mfma" not in k: continue
            e = acc[k[:90]][r['Counter_Name']]; e[0] += float(r['Counter_Value']); e[1] += 1
with open('$R/gpurun_out/pmc_layer_$L.txt', 'w') as out:
    for k, d in acc.items():
        print(k, file=out)
        for c, (s, n) in sorted(d.items()): print(f'   {c:40s} {s / n:16.1f}  ({n} launches)', file=out)
print(open('$R/gpurun_out/pmc_layer_$L.txt').read())
PY
