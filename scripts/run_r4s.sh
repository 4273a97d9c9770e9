#!/bin/bash
# round 4: status row behind the receive-buffer allocations (group suites), the default line with 4 index slots per entry
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4s; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
timeout 900 python -m pytest tests/test_gpu_bloom.py tests/test_gpu_exchange.py tests/test_gpu_rccl_transport.py tests/test_gpu_bench_multirank.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "$J
r=d['roofline']
print('default', d['value'], d['ms_per_step'], 'frac', r['frac'], 'alg', r['frac_algorithmic'], 'B/cert', r['traffic_measurement'] and r['traffic_measurement']['traffic_bytes_per_cert'], d['kernel_ms'], d['cpu_baseline']['value'], d['parity_vs_oracle_on_sample'], d['checks']['entries_disagreeing_with_generator'])
for k,v in d.get('secondary',{}).items(): print('  ', k, v['value'], v['ms_per_step'], v['map_ms'])" $OUT/bench_default.json; tail -2 $OUT/bench_default.err
timeout 600 python bench.py --raw --no-cpu --no-secondary --traffic off --steps 4 > $OUT/bench_raw.json 2> $OUT/bench_raw.err; python -c "$J
print('raw', d['value'], d['ms_per_step'], d['kernel_ms'])" $OUT/bench_raw.json
