#!/bin/bash
# PMC passes (no trace flags) over `bench.py --meta` to see what k_meta_new waits on
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_meta; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_ATOMIC_sum TCC_EA_ATOMIC_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $OUT/p$i -o p$i --output-format csv -- python $R/bench.py --meta --entries 20000000 --steps 2 --warmup 1 --no-cpu > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/fail.log
done
python $R/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
grep -E "k_meta_new|k_map_fused" $OUT/summary.txt
