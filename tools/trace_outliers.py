#!/usr/bin/env python3
"""Launches that take long on a tiny grid: from a rocprofv3 kernel trace (<dir>/**/*_kernel_trace.csv), every kernel
whose grid has at most `--wgs` workgroups and whose duration exceeds `--us` microseconds, grouped by name.  (The
one-workgroup halo copy inside the priority x update of round 3 -- 0.44 ms per step on the subdomain without an upper
neighbour -- is the kind of thing this finds.)

    python tools/trace_outliers.py gpurun_out/prof_r03_halo_2/stats [--wgs 8] [--us 30]
"""
import argparse
import csv
import glob
import os
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--wgs", type=int, default=8)
ap.add_argument("--us", type=float, default=30.0)
a = ap.parse_args()
acc = defaultdict(lambda: [0, 0.0, 0.0])
total = 0
for f in glob.glob(os.path.join(a.dir, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        total += 1
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        wgs = grid // max(wg, 1)
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if wgs <= a.wgs and us > a.us:
            e = acc[(r["Kernel_Name"][:90], wgs)]
            e[0] += 1
            e[1] += us
            e[2] = max(e[2], us)
print("%d launches in the trace; at most %d workgroups and longer than %.0f us:" % (total, a.wgs, a.us))
for (name, wgs), (cnt, tot, mx) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-90s wgs %3d  calls %5d  avg %8.1f us  max %8.1f us" % (name, wgs, cnt, tot / cnt, mx))
if not acc:
    print("(none)")
