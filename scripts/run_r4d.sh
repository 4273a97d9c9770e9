#!/bin/bash
# round 4: configs[4] whole — stream + write-back (tests, then disk at 4 M entries and noop at 200 M and 1 B)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4d; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream_writeback.py tests/test_gpu_exchange.py tests/test_gpu_rccl_transport.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -B5 -A25 "Error\|assert" $OUT/pytest.log | head -60
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
nproc; free -g | head -2; df -h /dev/shm | tail -1
timeout 900 python bench.py --stream 4000000 --entries 1000000 --write-back disk --traffic off > $OUT/stream_disk_4m.json 2> $OUT/stream_disk_4m.err; python -c "$J
print('disk 4M', d['value'], d['ms_per_step'], json.dumps(d['write_back'])[:1500])" $OUT/stream_disk_4m.json || tail -5 $OUT/stream_disk_4m.err
timeout 900 python bench.py --stream 200000000 --write-back noop --traffic off > $OUT/stream_noop_200m.json 2> $OUT/stream_noop_200m.err; python -c "$J
print('noop 200M', d['value'], d['ms_per_step'], json.dumps(d['write_back'])[:900], d['result'])" $OUT/stream_noop_200m.json || tail -5 $OUT/stream_noop_200m.err
timeout 1500 python bench.py --stream 1000000000 --write-back noop --traffic off > $OUT/stream_noop_1b.json 2> $OUT/stream_noop_1b.err; python -c "$J
print('noop 1B', d['value'], d['ms_per_step'], json.dumps(d['write_back'])[:900], d['result'])" $OUT/stream_noop_1b.json || tail -5 $OUT/stream_noop_1b.err
