cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "18 18 320 3" "18 36 160 3"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc1_$tag -o p -- python $R/scratch/one_conv.py $cfg > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc2_$tag -o p -- python $R/scratch/one_conv.py $cfg > /dev/null 2>&1
done
python - <<'PY'
import csv,glob,os
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/pmc*_*')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        rows=[r for r in csv.DictReader(open(f)) if 'conv_mfma' in r['Kernel_Name']]
        agg={}
        for r in rows:
            agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
        print(os.path.basename(d), {k: round(sum(v)/len(v)) for k,v in agg.items()}, 'vgpr', rows[0].get('VGPR_Count'), 'agpr', rows[0].get('Accum_VGPR_Count'), 'lds', rows[0].get('LDS_Block_Size'), 'grid', rows[0].get('Grid_Size'))
        os.remove(f)
PY
rm -rf $R/gpurun_out/pmc*_*
