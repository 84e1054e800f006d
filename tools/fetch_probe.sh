#!/bin/bash
# FETCH_SIZE of the plain SpMV for a few settings of an env var:
#   tools/fetch_probe.sh VAR "v1 v2 ..." [variants]
VAR=$1; VALUES=$2; VARIANTS=${3:-4}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in $VALUES; do
  rm -rf $ROOT/gpurun_out/fp_$v
  env $VAR=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/fp_$v -- python3 $ROOT/tools/spmv_probe.py --only-spmv --variants $VARIANTS --reps 5 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$ROOT/gpurun_out/fp_$v/**/*_counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-36:]
    if "spmv" not in k: continue
    a = acc.setdefault(k, [0.0, 0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (s, n, d) in acc.items():
    print("$VAR=$v", k, "reads GB %.3f" % (2*s/n*1024/1e9), "avg us %.1f" % (d/n/1e3))
PY
done
