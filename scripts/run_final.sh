#!/bin/bash
# what the driver does at round end (build check, smoke, GPU tests, default bench) + the profiled run of the same command
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT
cd $R
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_final.log 2>&1; tail -3 $OUT/pytest_final.log
timeout 900 python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_final.json').read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_physical'], d['kernel_ms'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['parity_vs_oracle_on_sample'])"; tail -2 $OUT/bench_final.err
( cd /tmp; export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o final --output-format csv -- python $R/bench.py --no-cpu > $OUT/bench_final_profiled.json 2> $OUT/bench_final_profiled.err
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/rocprofv3_kernel_stats_final.csv; head -8 "$f" | cut -c1-160
  find $OUT/prof -name "*.csv" -size +1M -delete )
python -c "
import json; d=json.loads([l for l in open('$OUT/bench_final_profiled.json').read().splitlines() if l.startswith('{')][-1]); print('profiled run:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms'])"
timeout 900 python bench.py --raw --meta --pem --no-cpu > $OUT/bench_raw_meta.json 2> $OUT/bench_raw_meta.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_raw_meta.json').read().splitlines() if l.startswith('{')][-1]); print('raw+meta+pem', d['value'], d['ms_per_step'], d['kernel_ms'], d.get('meta'), d.get('pem'))"
timeout 900 python bench.py --stream 1000000000 --no-cpu > $OUT/bench_stream.json 2> $OUT/bench_stream.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_stream.json').read().splitlines() if l.startswith('{')][-1]); print('stream', d['value'], d['ms_per_step'], d['result'])"
( cd /tmp; export TMPDIR=/tmp   # HBM bytes of k_meta_new (PMC pass of its own, no trace flags)
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_meta_fetch -o f --output-format csv -- python $R/bench.py --meta --entries 20000000 --steps 2 --warmup 1 --no-cpu > $OUT/pmc_meta_fetch.log 2>&1
  python $R/scripts/pmc_summary.py $OUT/pmc_meta_fetch 2>/dev/null | grep -E "k_meta_new" | tee $OUT/pmc_meta_fetch_summary.txt
  find $OUT/pmc_meta_fetch -name "*.csv" -size +1M -delete )
