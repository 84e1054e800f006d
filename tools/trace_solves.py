#!/usr/bin/env python3
"""Per-local-solve kernel summary from a rocprofv3 kernel trace (steady state: last 40 % of the
trace): one line per solve with (count, mean us) per kernel."""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.6):]
segs, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "spmv_pair_kernel<4" in n or "spmv_pair_kernel<2" in n or "spmv_pair_kernel<3" in n:
        cur = []
        segs.append(cur)
    if cur is not None:
        cur.append((n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for i, sg in enumerate(segs[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]):
    acc = defaultdict(list)
    for n, d in sg:
        acc[n.split("(")[0].replace("void ", "").replace("schwz::", "")[:34]].append(d)
    print("solve", i, "total %.0f us" % sum(d for _, d in sg),
          {k: (len(v), round(sum(v) / len(v), 1)) for k, v in acc.items()})
