#!/bin/bash
# what the driver does at round end (build check, smoke, GPU tests, default bench) + the profiled run of the same command
# and the secondary bench lines DESIGN.md quotes.  Round 4 set.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_final.log 2>&1; grep -E "passed|failed" $OUT/pytest_final.log | tail -2
timeout 900 python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err; python -c "$J
r=d['roofline']
print('default', d['value'], d['ms_per_step'], 'frac', r['frac'], 'alg', r['frac_algorithmic'], 'B/cert', r['traffic_measurement'] and r['traffic_measurement']['traffic_bytes_per_cert'], 'overfetch', r.get('over_fetch_vs_needed_lines'), d['kernel_ms'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['parity_vs_oracle_on_sample'], d['checks'])
for k,v in d.get('secondary',{}).items(): print('  ', k, {a:b for a,b in v.items() if a not in ('workload','note')})" $OUT/bench_final.json; tail -2 $OUT/bench_final.err
( cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o final --output-format csv -- python $R/bench.py --no-cpu --traffic off --no-secondary > $OUT/bench_final_profiled.json 2> $OUT/bench_final_profiled.err
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/rocprofv3_kernel_stats_final.csv; head -8 "$f" | cut -c1-160
  find $OUT/prof -name "*.csv" -size +1M -delete )
python -c "$J
print('profiled run:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms'])" $OUT/bench_final_profiled.json
[ -n "${CORE:-}" ] && exit 0
# N > 1 as the driver would start it, here with rank processes on the one GPU over the stand-in librccl (tests/harness)
python -c "from tests.harness import build_fake_rccl; print(build_fake_rccl())" > $OUT/fake.path
for spec in "2 bloom" "4 bloom" "4 owner" "2 local"; do
  set -- $spec
  CTMR_RCCL_LIB=$(cat $OUT/fake.path) timeout 900 python bench.py --gpus $1 --dedup $2 --total-entries 16000000 --steps 3 --traffic off > $OUT/bench_fake_rccl_n$1_$2.json 2> $OUT/bench_fake_rccl_n$1_$2.err; python -c "$J
print('gpus $1 $2 (one GPU, stand-in librccl)', d['value'], d['ms_per_step'], d['scaling'], d['checks'], d['parity_vs_oracle_on_sample'], d['exchange']['ms_phase_rank0'], d['exchange']['wire_bytes_sent_by_rank0_per_step'])" $OUT/bench_fake_rccl_n$1_$2.json || tail -5 $OUT/bench_fake_rccl_n$1_$2.err
done
CTMR_RCCL_LIB=$(cat $OUT/fake.path) timeout 600 python bench.py --gpus 3 --stream 48000000 --entries 12000000 --traffic off > $OUT/bench_fake_rccl_n3_stream.json 2> $OUT/bench_fake_rccl_n3_stream.err; python -c "$J
print('gpus 3 stream (one GPU, stand-in librccl)', d['value'], d['ms_per_step'], d['config']['dedup'], d['result'], d['exchange'])" $OUT/bench_fake_rccl_n3_stream.json || tail -5 $OUT/bench_fake_rccl_n3_stream.err
timeout 600 python bench.py --strict-strings --no-cpu --no-secondary --traffic off > $OUT/bench_strictstrings.json 2> $OUT/bench_strictstrings.err; python -c "$J
print('--strict-strings', d['value'], d['ms_per_step'], d['kernel_ms'])" $OUT/bench_strictstrings.json
for m in "--raw" "--raw --meta --pem" "--meta" "--stream 1000000000" "--global-dedup owner" "--global-dedup bloom" "--raw --trusted-chain"; do
  tag=$(echo $m | tr -d ' -'); timeout 900 python bench.py $m --no-cpu --steps 3 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python -c "$J
r=d['roofline']
print('$m', d['value'], d['ms_per_step'], 'frac', r['frac'], 'alg', r.get('frac_algorithmic'), r.get('invalid'), d.get('kernel_ms'), d.get('exchange',{}).get('ms_phase_rank0'), (d.get('roofline_decode_match') or {}).get('frac'), d.get('result',{}).get('duplicate_structure_matches_generator_in_every_wave'))" $OUT/bench_$tag.json || tail -3 $OUT/bench_$tag.err
done
for spec in "50000000 2" "25000000 4" "12500000 8"; do
  set -- $spec
  timeout 600 python scripts/rank_cost_at_world.py $1 $2 > $OUT/rank_cost_$1_w$2.json 2>&1; tail -1 $OUT/rank_cost_$1_w$2.json | cut -c1-600
done
