#!/bin/bash
# kernel-time breakdown of subdomains WITH neighbours: tools/halo_probe.py P (slabs of 512 x 512 x 64 in one process)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
P=${1:-2}
OUT=$ROOT/gpurun_out/prof_r03_halo_$P
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/halo_probe.py $P > $OUT/stats.log 2>&1 || exit 1
f=$(find $OUT/stats -name "*_kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-86s calls %5s avg %9.1f us  %5s%%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -2 $OUT/stats.log
python3 $ROOT/tools/trace_outliers.py $OUT/stats
