#!/bin/bash
# the whole Store() semantic on raw get-entries buffers: decode + Chain[0] match + map/reduce + IssuerMetadata memo + PEM
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_entries.py tests/test_gpu_meta.py tests/test_gpu_pem.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --raw --meta --pem --steps 3 --warmup 1 --no-cpu > $OUT/bench_raw_meta_pem_40m.json 2> $OUT/bench_raw_meta_pem_40m.err; python -c "
import json; d=json.load(open('$OUT/bench_raw_meta_pem_40m.json')); print('raw', d['value'], d['kernel_ms']); print(d.get('meta')); print(d.get('pem'))"; tail -2 $OUT/bench_raw_meta_pem_40m.err
