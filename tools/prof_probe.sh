#!/bin/bash
# per-kernel times of the CG loop of tools/spmv_probe.py for a grid shape:  tools/prof_probe.sh 512,512,64
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SHAPE=${1:-256,256,256}
OUT=$ROOT/gpurun_out/prof_probe
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/spmv_probe.py --shape $SHAPE --variants 0 --reps 20 > $OUT/log.txt 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print("%-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
