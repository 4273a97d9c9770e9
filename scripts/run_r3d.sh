#!/bin/bash
# round 3: LOCAL-group concurrency trace, stream line priced, multi-rank bench after the setup change
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3d; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bench_multirank.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -o lg --output-format csv -- python $R/scripts/local_group_overlap.py run > $OUT/local_group_run.txt 2>&1 )
python scripts/local_group_overlap.py report $OUT/trace > $OUT/local_group_kernel_overlap.txt 2>&1; tail -8 $OUT/local_group_kernel_overlap.txt
find $OUT/trace -name "*.csv" -size +2M -delete
timeout 900 python bench.py --stream 1000000000 --no-cpu > $OUT/bench_stream.json 2> $OUT/bench_stream.err; tail -2 $OUT/bench_stream.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_stream.json').read().splitlines() if l.startswith('{')][-1]); r=d['roofline']; print('stream', d['value'], d['ms_per_step'], r['frac'], r.get('frac_algorithmic'), r.get('traffic_measurement',{}).get('traffic_bytes_per_cert'), d['result'])"
