#!/bin/bash
# in-box A/B of the solve start: chunk-by-chunk start launch + kSpmvDotSym + full last direction update
# (round-2 state) against the z-sweep start (INIT / FIRST forms) and the state-only last iteration
set -o pipefail
ARGS="--steps 30 --warmup 3 --no-cpu-baseline --no-ttr --no-plain-loop $BENCH_ARGS"
for rep in 1 2; do
  for cfg in "0 1" "0 0" "1 1" "1 0"; do
    set -- $cfg
    echo "SWEEPSTART=$1 LASTDIR=$2: $(SCHWZ_CG_SWEEPSTART=$1 SCHWZ_CG_LASTDIR=$2 python3 bench.py $ARGS 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4f ms/step  %.1f iter/s  upd %.4f ms  dirdot %.4f ms" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["roofline_spmv"]["avg_launch_ms"]))')"
  done
done
