#!/bin/bash
# round 4: strict_strings inside the walk; the exact command the driver issues at N = 8 over the stand-in librccl; fuzz campaigns with key mutations
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4e; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
for m in "" "--strict-strings" "" "--strict-strings"; do
  tag=$(echo "d$m" | tr -d ' -')
  timeout 600 python bench.py $m --no-cpu --no-secondary --traffic off --steps 8 --warmup 2 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python -c "$J
print('$m', d['value'], d['ms_per_step'], d['kernel_ms'])" $OUT/bench_$tag.json || tail -3 $OUT/bench_$tag.err
done
python -c "from tests.harness import build_fake_rccl; print(build_fake_rccl())" > $OUT/fake.path
CTMR_RCCL_LIB=$(cat $OUT/fake.path) timeout 1200 python bench.py --gpus 8 --total-entries 100000000 --steps 3 --warmup 1 --traffic off > $OUT/bench_fake_rccl_n8_100m.json 2> $OUT/bench_fake_rccl_n8_100m.err; python -c "$J
print('gpus 8 (one GPU, stand-in librccl)', d['value'], d['ms_per_step'], d['scaling'], d['config'].get('dedup'), d['checks'], d['parity_vs_oracle_on_sample'], d['exchange']['ms_phase_rank0'], d['exchange']['wire_bytes_sent_by_rank0_per_step'])" $OUT/bench_fake_rccl_n8_100m.json || tail -8 $OUT/bench_fake_rccl_n8_100m.err
timeout 900 python scripts/fuzz_gpu.py ${FUZZ_N:-20000000} 20260925 > $OUT/fuzz_gpu_certificates_keys.txt 2>&1; tail -2 $OUT/fuzz_gpu_certificates_keys.txt
STRICT_STRINGS=1 timeout 600 python scripts/fuzz_gpu.py 5000000 20260926 > $OUT/fuzz_gpu_certificates_strict_strings.txt 2>&1; tail -2 $OUT/fuzz_gpu_certificates_strict_strings.txt
STRICT_LEAF=1 timeout 600 python scripts/fuzz_gpu_entries.py 4000000 7101 > $OUT/fuzz_gpu_entries_strict_leaf.txt 2>&1; tail -2 $OUT/fuzz_gpu_entries_strict_leaf.txt
timeout 600 python scripts/fuzz_gpu_groups.py 1500 > $OUT/fuzz_gpu_groups.txt 2>&1; tail -2 $OUT/fuzz_gpu_groups.txt
