#!/bin/bash
# what the driver does at round end (build check, smoke, GPU tests, default bench) + the profiled run of the same command
# and the secondary bench lines DESIGN.md quotes
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_final.log 2>&1; grep -E "passed|failed" $OUT/pytest_final.log | tail -2
timeout 900 python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_final.json').read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('default', d['value'], d['ms_per_step'], 'frac', r['frac'], 'alg', r['frac_algorithmic'], 'B/cert', r['traffic_measurement'] and r['traffic_measurement']['traffic_bytes_per_cert'], 'overfetch', r.get('over_fetch_vs_needed_lines'), d['kernel_ms'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['parity_vs_oracle_on_sample'], d['parity_sample'])"; tail -2 $OUT/bench_final.err
( cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o final --output-format csv -- python $R/bench.py --no-cpu --traffic off > $OUT/bench_final_profiled.json 2> $OUT/bench_final_profiled.err
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/rocprofv3_kernel_stats_final.csv; head -8 "$f" | cut -c1-160
  find $OUT/prof -name "*.csv" -size +1M -delete )
python -c "
import json; d=json.loads([l for l in open('$OUT/bench_final_profiled.json').read().splitlines() if l.startswith('{')][-1]); print('profiled run:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms'])"
[ -n "${CORE:-}" ] && exit 0
timeout 900 python bench.py --mixed --no-cpu > $OUT/bench_mixed.json 2> $OUT/bench_mixed.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_mixed.json').read().splitlines() if l.startswith('{')][-1]); print('mixed', d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'])"
timeout 900 python bench.py --raw --meta --pem --no-cpu > $OUT/bench_raw_meta.json 2> $OUT/bench_raw_meta.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_raw_meta.json').read().splitlines() if l.startswith('{')][-1]); print('raw+meta+pem', d['value'], d['ms_per_step'], d['kernel_ms'], d.get('meta'), d.get('pem'))"
timeout 900 python bench.py --raw --no-cpu > $OUT/bench_raw.json 2> $OUT/bench_raw.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_raw.json').read().splitlines() if l.startswith('{')][-1]); print('raw', d['value'], d['ms_per_step'], d['kernel_ms'])"
timeout 900 python bench.py --stream 1000000000 --no-cpu > $OUT/bench_stream.json 2> $OUT/bench_stream.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_stream.json').read().splitlines() if l.startswith('{')][-1]); print('stream', d['value'], d['ms_per_step'], d['result'])"
for m in owner bloom; do
  timeout 600 python bench.py --global-dedup $m --no-cpu --steps 3 > $OUT/bench_gd_$m.json 2> $OUT/bench_gd_$m.err; python -c "
import json; d=json.load(open('$OUT/bench_gd_$m.json')); print('$m', d['value'], d['ms_per_step'], d['result']['global_dedup'])"
done
# one rank under the launcher, as the driver starts N > 1: the RCCL group path of bench.py with a world of one
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --traffic off --entries 20000000 > $OUT/bench_launcher_world1.json 2> $OUT/bench_launcher_world1.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_launcher_world1.json').read().splitlines() if l.startswith('{')][-1]); print('launcher world 1', d['value'], d['n_gpus'], d['config']['parallelism'], d['result'])"; tail -3 $OUT/bench_launcher_world1.err
