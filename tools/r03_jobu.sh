#!/bin/bash
# kernel-time breakdown of the plain-CSR whole step (general-matrix path); extra env settings as arguments,
# one profile per argument ("-" = defaults)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
[ $# -eq 0 ] && set -- -
n=0
for setting in "$@"; do
  n=$((n + 1))
  OUT=$ROOT/gpurun_out/prof_r03_plainloop_$n
  rm -rf $OUT && mkdir -p $OUT
  [ "$setting" != "-" ] && export $setting
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ttr --no-plain-loop --no-mirror --no-shapes --strong-grid= --spmv-variant 6 > $OUT/stats.log 2>&1 || exit 1
  [ "$setting" != "-" ] && unset ${setting%%=*}
  f=$(find $OUT/stats -name "*_kernel_stats.csv" | head -1)
  echo "== $setting" >> $ROOT/gpurun_out/r03_plainloop_kernels.txt
  python3 - $f >> $ROOT/gpurun_out/r03_plainloop_kernels.txt <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    if "stream_copy" in r["Name"] or "stream_read" in r["Name"] or "rocclr" in r["Name"]:
        continue
    print("%-70s calls %5s avg %9.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
cat $ROOT/gpurun_out/r03_plainloop_kernels.txt
