#!/bin/bash
# IssuerMetadata memo kernel: parity tests, then bench --meta (packed 100 M) and --raw --meta (40 M)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/meta; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_meta.py tests/test_storage_gpu.py tests/test_host_cpp.py -x -q -m gpu > $OUT/pytest_meta.txt 2>&1; tail -4 $OUT/pytest_meta.txt
timeout 900 python bench.py --meta --no-cpu > $OUT/bench_meta.json 2> $OUT/bench_meta.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_meta.json').read().splitlines() if l.startswith('{')][-1]); print('meta', d['value'], d['ms_per_step'], d['kernel_ms'], d.get('meta'))"
timeout 900 python bench.py --raw --meta --no-cpu > $OUT/bench_raw_meta.json 2> $OUT/bench_raw_meta.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_raw_meta.json').read().splitlines() if l.startswith('{')][-1]); print('raw+meta', d['value'], d['ms_per_step'], d['kernel_ms'])"
