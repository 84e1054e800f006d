#!/bin/bash
# plain-CSR whole step: settings A/B, interleaved repeats on one box (arguments: env settings, "-" = defaults)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03_plainloop_ab.txt
: > $OUT
REPS=${REPS:-4}
[ $# -eq 0 ] && set -- -
for rep in $(seq $REPS); do
  for setting in "$@"; do
    [ "$setting" != "-" ] && export $setting
    line=$(python3 $ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-ttr --no-plain-loop --no-mirror --no-shapes --strong-grid= --spmv-variant 6 2>/dev/null | tail -1) || exit 1
    [ "$setting" != "-" ] && unset ${setting%%=*}
    python3 - "$setting" "$line" >> $OUT <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print("%-34s value %.1f  ms/step %.4f" % (sys.argv[1], d["value"], d["ms_per_step"]))
PY
  done
done
sort $OUT
