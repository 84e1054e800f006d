#!/usr/bin/env python3
"""GPU idle time between consecutive kernels from a rocprofv3 --kernel-trace CSV:
    python tools/trace_gaps.py gpurun_out/prof_xx/stats [last_n_kernels]
"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 700
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-last:]
gaps = collections.defaultdict(list)
busy = 0
for a, b in zip(rows[:-1], rows[1:]):
    gaps[(a["Kernel_Name"][:44], b["Kernel_Name"][:44])].append(
        (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for r in rows:
    busy += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("kernels %d  span %.3f ms  busy %.3f ms  idle %.1f%%" % (len(rows), span / 1e6, busy / 1e6,
                                                             100.0 * (span - busy) / span))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("%-46s -> %-46s n=%4d avg %7.1f us max %8.1f total %8.1f" % (
        k[0], k[1], len(v), sum(v) / len(v), max(v), sum(v)))
