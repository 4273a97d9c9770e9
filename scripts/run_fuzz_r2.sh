#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/fuzz_r2; mkdir -p $OUT
cd $R
timeout 1500 python scripts/fuzz_gpu.py ${1:-3000000} ${2:-20260925} > $OUT/fuzz_gpu_certificates.txt 2>&1; echo "rc $?" >> $OUT/fuzz_gpu_certificates.txt; tail -3 $OUT/fuzz_gpu_certificates.txt
timeout 900 python scripts/fuzz_gpu_entries.py ${3:-1000000} > $OUT/fuzz_gpu_entries.txt 2>&1; echo "rc $?" >> $OUT/fuzz_gpu_entries.txt; tail -3 $OUT/fuzz_gpu_entries.txt
