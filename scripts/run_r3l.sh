#!/bin/bash
# round 3: how the 64-byte slot image leaves the map kernel — four lanes per slot (shipped), every claiming lane its own 4 x 16 B at
# the end of the kernel (LANE_STORE), or at once when the CAS succeeded (STORE_AT_CLAIM); A/B/C twice on one box; checks on
mkdir -p gpurun_out/r3l
for rep in 1 2; do
  for lib in libctmr.so libctmr_LANE_STORE.so libctmr_STORE_AT_CLAIM.so; do
    CTMR_LIB=$PWD/ct_mapreduce_amd/$lib timeout 200 python bench.py --no-cpu --no-secondary --traffic off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['kernel_ms'], d['checks']['entries_disagreeing_with_generator'])" | tee -a gpurun_out/r3l/ab_slot_store.txt
  done
done
