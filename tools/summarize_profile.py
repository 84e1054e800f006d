#!/usr/bin/env python3
"""Turns the rocprofv3 passes written by tools/profile_bench.sh into the files
committed under profiles/:

    python tools/summarize_profile.py gpurun_out/prof_r01 r01

  profiles/<tag>_kernel_stats.csv   copy of rocprofv3 --stats per-kernel summary
  profiles/<tag>_summary.json       per kernel: calls, avg ns, FETCH_SIZE/WRITE_SIZE per launch
  profiles/traffic.json             HBM bytes per launch of the dominant kernel (read by bench.py)

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and
WRITE_SIZE are collected in separate passes and are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact
for 16-byte-per-lane streaming stores.
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys
from collections import defaultdict

TRACKED = ("spmv_stream_kernel<0,", "spmv_pair_sweep_kernel<", "spmv_pair_dirdot_sweep_kernel<", "spmv_pair_kernel<8,", "spmv_pair_kernel<6,", "spmv_pair_kernel<7,", "spmv_pair_kernel<5,", "spmv_pair_kernel<1,", "spmv_pattern_kernel<1,",
           "spmv_dict_kernel<1>", "spmv_tiled2_kernel<1>", "spmv_stream_kernel<1,", "cg_update_kernel", "cg_direction_kernel")


def kernel_source_hash(root):
    """The same hash bench.py computes: traffic figures are only quoted for the kernel sources
    they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(root, "schwarz-lib_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):  # device code (host_setup.cpp is host-only)
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def read_counter(dirname, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items() if v[1]}


def plain(prefix, tag):
    """--plain <gpurun_out/prof_<tag>_plain> <tag>: the per-shape passes of tools/profile_plain.sh -> one
    profiles/<tag>_plain_<shape>_kernel_stats.csv per shape and the spmv_stream_kernel<0, entry of that shape
    in profiles/traffic.json (the other entries of the shape are kept)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    tpath = os.path.join(out, "traffic.json")
    allent = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for d in sorted(glob.glob(prefix + "_*")):
        key = os.path.basename(d).split("_plain_")[-1]
        stats = glob.glob(os.path.join(d, "stats", "**", "*_kernel_stats.csv"), recursive=True)
        if not stats:
            continue
        shutil.copy(stats[0], os.path.join(out, "%s_plain_%s_kernel_stats.csv" % (tag, key)))
        fetch = read_counter(os.path.join(d, "fetch"), "FETCH_SIZE")
        write = read_counter(os.path.join(d, "write"), "WRITE_SIZE")
        for row in csv.DictReader(open(stats[0])):
            name = row["Name"]
            if "spmv_stream_kernel<0," not in name or name not in fetch or name not in write:
                continue
            ent = allent.setdefault(key, {})
            ent["spmv_stream_kernel<0,"] = dict(
                hbm_bytes_per_launch=(2.0 * fetch[name] + write[name]) * 1024.0, avg_ns=float(row["AverageNs"]),
                fetch_size_kib_raw=fetch[name], write_size_kib=write[name], calls=int(row["Calls"]),
                source="%s_plain_%s_kernel_stats.csv" % (tag, key),
                note="(2*FETCH_SIZE + WRITE_SIZE) KiB per launch, separate --pmc passes (tools/profile_plain.sh)")
            ent["kernel_source_hash"] = kernel_source_hash(root)
            print("%-16s %-50s avg %8.1f us  hbm %.3f GB" % (key, name[:50], float(row["AverageNs"]) / 1e3,
                                                          ent["spmv_stream_kernel<0,"]["hbm_bytes_per_launch"] / 1e9))
    json.dump(allent, open(tpath, "w"), indent=1, sort_keys=True)


def main():
    if sys.argv[1] == "--plain":
        return plain(sys.argv[2], sys.argv[3])
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(out, tag + "_kernel_stats.csv"))
    fetch = read_counter(os.path.join(src, "fetch"), "FETCH_SIZE")
    write = read_counter(os.path.join(src, "write"), "WRITE_SIZE")
    summary = {}
    for row in csv.DictReader(open(stats)):
        name = row["Name"]
        f_kib, w_kib = fetch.get(name), write.get(name)
        entry = dict(calls=int(row["Calls"]), avg_ns=float(row["AverageNs"]),
                     pct=float(row["Percentage"]),
                     fetch_size_kib_raw=f_kib, write_size_kib=w_kib)
        if f_kib is not None and w_kib is not None:
            entry["hbm_bytes_per_launch"] = (2.0 * f_kib + w_kib) * 1024.0
            entry["hbm_GBs"] = entry["hbm_bytes_per_launch"] / entry["avg_ns"]
        summary[name] = entry
    json.dump(summary, open(os.path.join(out, tag + "_summary.json"), "w"), indent=1, sort_keys=True)
    # HBM bytes per launch of the CG kernels, keyed by the name prefix bench.py looks up
    tpath = os.path.join(out, "traffic.json")
    key = sys.argv[3] if len(sys.argv) > 3 else "256x256x256"
    ent = {}
    for name in TRACKED:
        hit = [v for k, v in summary.items() if name in k and "hbm_bytes_per_launch" in v]
        # several instantiations share a prefix (the INIT / FIRST forms of the walk run once per solve): the
        # one launched most is the one bench.py times
        hit.sort(key=lambda v: -v["calls"])
        if hit:
            ent[name] = dict(hbm_bytes_per_launch=hit[0]["hbm_bytes_per_launch"], avg_ns=hit[0]["avg_ns"],
                             source=tag + "_summary.json",
                             note="(2*FETCH_SIZE + WRITE_SIZE) KiB per launch, separate --pmc passes")
    if ent:
        ent["kernel_source_hash"] = kernel_source_hash(root)
        allent = {}
        if os.path.exists(tpath):  # one entry per grid shape: keep the others
            try:
                allent = json.load(open(tpath))
            except Exception:
                allent = {}
        allent[key] = ent
        json.dump(allent, open(tpath, "w"), indent=1, sort_keys=True)
    for k, v in sorted(summary.items(), key=lambda kv: -kv[1]["pct"])[:8]:
        print("%-60s calls %4d avg %9.1f us  hbm %s" % (
            k[:60], v["calls"], v["avg_ns"] / 1e3,
            ("%.3f GB (%.0f GB/s)" % (v["hbm_bytes_per_launch"] / 1e9, v["hbm_GBs"]))
            if "hbm_bytes_per_launch" in v else "n/a"))


if __name__ == "__main__":
    main()
