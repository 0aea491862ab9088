#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats, FETCH_SIZE / WRITE_SIZE passes) into the small
files kept under profiles/.  Usage:
    tools/summarize_rocprof.py <kernel-trace dir> <fetch dir> <write dir> <out prefix>
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled for wide coalesced reads when comparing
with byte counts (MI355X_MICROARCH.md, HBM section) -- the raw value is stored here."""
import collections
import csv
import glob
import json
import re
import shutil
import sys


def short(name):
    m = re.search(r"([a-z_0-9]+_kernel)", name)
    k = m.group(1) if m else name[:40]
    if k == "spectral_kernel":
        k = "spectral_kernel(border list)"
    return k


def main():
    kt, fetch, write, out = sys.argv[1:5]
    stats = (glob.glob(kt + "/*/*kernel_stats.csv") + glob.glob(kt + "/*kernel_stats.csv"))[0]
    shutil.copy(stats, out + "_kernel_stats.csv")
    res = {}
    for d in (fetch, write):
        f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), r["Counter_Name"])
            agg[key][0] += float(r["Counter_Value"])
            agg[key][1] += 1
        for (k, c), (v, n) in agg.items():
            res.setdefault(k, {})[c + "_KB_total"] = v
            res[k]["launches"] = n
            res[k][c + "_GB_per_launch"] = round(v * 1024 / n / 1e9, 4)
    json.dump(res, open(out + "_pmc_fetch_write.json", "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_KB_total", 0))[:14]:
        print(f"{k:34s} launches {v['launches']:4d} fetch/launch "
              f"{v.get('FETCH_SIZE_GB_per_launch', 0):8.3f} GB  write/launch "
              f"{v.get('WRITE_SIZE_GB_per_launch', 0):8.3f} GB")


if __name__ == "__main__":
    main()
