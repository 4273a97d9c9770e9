#!/bin/bash
# round 3, strict_strings: the new tests, then A/B/A/B of the default bench with and without the nf_extra pointer in the map kernel
mkdir -p gpurun_out/r3i
timeout 300 python -m pytest tests/test_gpu_strings.py tests/test_gpu_entries.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider --timeout 100 2>&1 | tail -12 | cut -c1-300 | tee gpurun_out/r3i/pytest.txt
for rep in 1 2; do
  for lib in libctmr.so libctmr_nonfx.so; do
    CTMR_LIB=$PWD/ct_mapreduce_amd/$lib timeout 200 python bench.py --no-cpu --no-secondary --traffic off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['kernel_ms'])" | tee -a gpurun_out/r3i/ab_nfx.txt
  done
done
