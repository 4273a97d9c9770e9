#!/bin/bash
# The one runner for a gpurun call (round 5: replaces the sixty one-off run_*.sh of rounds 1-4).
#   gpurun --timeout T -- 'bash scripts/run.sh TAG STEP [STEP …]'
# Every step writes into gpurun_out/TAG/ (merged back by gpurun) and prints one summary line.  Steps:
#   pytest[:EXPR]            python -m pytest tests -m gpu -x -q [-k EXPR]      (pytestall[:EXPR]: without -x, lists every failure)
#   bench:NAME:ARGS…         python bench.py ARGS…            → NAME.json (+ .err); ARGS separated by ':' or ','
#   lib:PATH                 export CTMR_LIB=PATH for the steps that follow (a sweep / experiment build); lib: resets it
#   prof:NAME:ARGS…          rocprofv3 --kernel-trace --stats of bench.py ARGS… → NAME_kernel_stats.csv
#   pmc:NAME:COUNTERS:ARGS…  rocprofv3 --pmc passes of bench.py ARGS… (',' separates passes, '+' joins the counters of one pass)
#                            → NAME_pmc.txt (per kernel means)
#   py:NAME:SCRIPT:ARGS…     python SCRIPT ARGS…              → NAME.txt      (fuzz campaigns, calibrations)
#   env:K=V                  export K=V for the steps that follow
#   hip:NAME:SRC:ARGS…       hipcc SRC (scripts/*.hip) && run it with ARGS → NAME.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
J='import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
except Exception as e:
    print("no JSON line:", e); sys.exit(0)
r=d.get("roofline",{})
print({k:d.get(k) for k in ("value","ms_per_step","n_gpus")}, "map_ms", d.get("kernel_ms",{}).get("map"), "frac", r.get("frac"), "alg", r.get("frac_algorithmic"),
      "checks", d.get("checks"), "parity", d.get("parity_vs_oracle_on_sample"), (d.get("parity_sample") or {}).get("entries_checked_all_ranks"))
for k,v in (d.get("secondary") or {}).items(): print("  secondary", k, {x:v.get(x) for x in ("value","ms_per_step","map_ms","frac","traffic_bytes_per_cert","error","same_results_as_the_reference_profile")})
for k in ("pem","stream","write_back","exchange"):
    if k in d: print("  ",k,d[k])'
split() { echo "$1" | tr ':,' '  '; }
for STEP in "$@"; do
  KIND=${STEP%%:*}; REST=${STEP#*:}; [ "$REST" = "$STEP" ] && REST=""
  NAME=${REST%%:*}; ARGS=${REST#*:}; [ "$ARGS" = "$REST" ] && ARGS=""
  echo "== $STEP"
  case $KIND in
    env) export "$REST" ;;
    lib) if [ -n "$REST" ]; then export CTMR_LIB=$R/$REST; else unset CTMR_LIB; fi ;;
    pytest) (cd $R && timeout 1500 python -m pytest tests -m gpu -x -q ${REST:+-k "$REST"} > $OUT/pytest.txt 2>&1); grep -E "passed|failed|error" $OUT/pytest.txt | tail -3 ;;
    pytestall) (cd $R && timeout 2400 python -m pytest tests -m gpu -q ${REST:+-k "$REST"} > $OUT/pytest.txt 2>&1); grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.txt | tail -40 ;;
    bench) (cd $R && timeout 1500 python bench.py $(split "$ARGS") > $OUT/$NAME.json 2> $OUT/$NAME.err); python -c "$J" $OUT/$NAME.json; tail -2 $OUT/$NAME.err ;;
    prof) timeout 1500 rocprofv3 --kernel-trace --stats -d $OUT/$NAME.kt -o kt --output-format csv -- python $R/bench.py $(split "$ARGS") > $OUT/$NAME.kt.log 2>&1
          find $OUT/$NAME.kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${NAME}_kernel_stats.csv; rm -rf $OUT/$NAME.kt
          head -12 $OUT/${NAME}_kernel_stats.csv | cut -c1-180; python -c "$J" $OUT/$NAME.kt.log ;;
    pmc) CNT=${ARGS%%:*}; BARGS=${ARGS#*:}
         for c in $(echo $CNT | tr ',' ' '); do   # ',' separates passes, '+' joins counters of one pass (SQ: 8 slots, TCC: 4)
           timeout 1500 rocprofv3 --pmc $(echo $c | tr '+' ' ') -d $OUT/$NAME.pmc/$c -o pmc --output-format csv -- python $R/bench.py $(split "$BARGS") > $OUT/$NAME.$c.log 2>&1
         done
         python $R/scripts/pmc_summary.py $OUT/$NAME.pmc > $OUT/${NAME}_pmc.txt 2>&1; rm -rf $OUT/$NAME.pmc; head -40 $OUT/${NAME}_pmc.txt ;;
    py) SCRIPT=${ARGS%%:*}; PARGS=${ARGS#*:}; [ "$PARGS" = "$ARGS" ] && PARGS=""
        (cd $R && timeout 3000 python $SCRIPT $(split "$PARGS") > $OUT/$NAME.txt 2>&1); tail -4 $OUT/$NAME.txt ;;
    hip) SRC=${ARGS%%:*}; HARGS=${ARGS#*:}; [ "$HARGS" = "$ARGS" ] && HARGS=""
         /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/$SRC -o /tmp/$NAME.bin > $OUT/$NAME.txt 2>&1 && timeout 900 /tmp/$NAME.bin $(split "$HARGS") >> $OUT/$NAME.txt 2>&1; tail -30 $OUT/$NAME.txt ;;
    *) echo "unknown step $KIND" ;;
  esac
done
