#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the Infinity Cache order probe: do its launches fetch less than their bytes?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_icprobe
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SZ=${1:-16.9}
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $ROOT/tools/probes/ic_order_probe $SZ > $OUT/fetch.log 2>&1 || exit 2
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $ROOT/tools/probes/ic_order_probe $SZ > $OUT/write.log 2>&1 || exit 3
python3 - $OUT $SZ <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out, sz = sys.argv[1], float(sys.argv[2])
def rd(d, c):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                a = acc[r["Kernel_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}
f, w = rd("fetch", "FETCH_SIZE"), rd("write", "WRITE_SIZE")
n = sz * 1e6
for k in sorted(f):
    print("%-40s FETCH_SIZE %9.0f KiB (x2 = %.3f GB; the launch reads %.3f GB)  WRITE_SIZE %9.0f KiB (%.3f GB; writes %.3f GB)" % (
        k[:40], f[k], 2 * f[k] * 1024 / 1e9, 16 * n / 1e9, w.get(k, 0), w.get(k, 0) * 1024 / 1e9, 8 * n / 1e9))
PY
