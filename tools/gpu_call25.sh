#!/bin/bash
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out/sq_x3; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/tools/conv_microbench.py 64 112 224 256 256 3 1 x3 6"
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $OUT/$tag -o p --output-format csv -- $CMD > $OUT/$tag.log 2>&1
done
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda:[0,0])
for f in glob.glob('/root/repo/gpurun_out/sq_x3/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm_glds' in r['Kernel_Name']:
            a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for k,(v,n) in sorted(acc.items()): print(f"{k:32s} {v/n:16.0f}  (x{n})")
PY
find $OUT -name "*.csv" -size +2M -delete
