#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu8.log 2>&1; tail -6 $OUT/pytest_gpu8.log
timeout 600 python bench.py --raw --steps 3 --warmup 1 --no-cpu > $OUT/bench_raw_40m_d.json 2> $OUT/bench_raw_40m_d.err; python -c "
import json; d=json.load(open('$OUT/bench_raw_40m_d.json')); print('raw', d['value'], d['kernel_ms'])"
timeout 600 python scripts/host_batch_latency.py > $OUT/host_batch_latency2.json 2> $OUT/host_batch_latency2.err; cat $OUT/host_batch_latency2.json; tail -2 $OUT/host_batch_latency2.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_gd -o kt --output-format csv -- python $R/bench.py --global-dedup --entries 50000000 --steps 3 --warmup 1 --no-cpu > $OUT/kt_gd.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +1M -delete
find $OUT/kt_gd -name "*kernel_stats.csv" | head -1 | xargs head -16 | cut -c1-150
