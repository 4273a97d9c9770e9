#!/bin/bash
# usage: prof_pmc.sh OUTNAME "sweep cfgs" ENTRIES   — separate --pmc passes (no trace flags)
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
CFG=$2; E=$3
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $OUT/p$i -o p$i --output-format csv -- python $R/scripts/sweep.py $E "$CFG" > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/fail.log
done
python $R/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
