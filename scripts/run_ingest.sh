#!/bin/bash
# C++-driven ingestion rates of the host-buffer entry points (tests/host/ingest_bench.cpp)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ingest; mkdir -p $OUT
cd $R
g++ -std=c++17 -O2 -I include tests/host/ingest_bench.cpp -o /tmp/ingest_bench -L ct_mapreduce_amd -lctmr -Wl,-rpath,$R/ct_mapreduce_amd || exit 1
for ns in ${STREAMS:-2}; do
  CTMR_PIPE_STREAMS=$ns timeout 300 /tmp/ingest_bench 1001 6000 > $OUT/ingest_1001_s$ns.json 2> $OUT/ingest_1001_s$ns.err; echo "streams $ns"; python3 -c "
import json; d=json.load(open('$OUT/ingest_1001_s$ns.json'))
for r in d['results']: print(r['entry_point'], r['payload_memory'], r.get('tickets_in_flight'), round(r['certs_per_s']/1e6,2), 'M/s', r['payload_GBps'], 'GB/s')"
done
timeout 300 /tmp/ingest_bench 16384 400 > $OUT/ingest_16384.json 2> $OUT/ingest_16384.err; cat $OUT/ingest_16384.json
# RAW get-entries responses (ctmr_map_entries against ctmr_submit_entries / ctmr_wait_entries)
timeout 300 /tmp/ingest_bench 1001 3000 raw > $OUT/ingest_raw_1001.json 2> $OUT/ingest_raw_1001.err; tail -2 $OUT/ingest_raw_1001.err; python3 -c "
import json; d=json.load(open('$OUT/ingest_raw_1001.json'))
for r in d['results']: print('raw', r['entry_point'], r['payload_memory'], r.get('tickets_in_flight'), round(r['entries_per_s']/1e6,2), 'M entries/s', r['payload_GBps'], 'GB/s')"
