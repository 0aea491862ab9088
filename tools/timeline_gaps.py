"""Busy time against wall span of the greedy-PCA iterations in a rocprofv3 kernel trace.

usage: python tools/timeline_gaps.py <kernel_trace.csv | results.db>
An iteration starts at each pca_select*_kernel; for the last full PCA run in the trace the
script prints, per iteration, the wall span, the summed kernel time and the per-kernel split.
"""
import csv
import sys
from collections import defaultdict

rows = []


def clean(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]


if sys.argv[1].endswith(".db"):  # rocprofv3's rocpd output (ROCm 7.2 default)
    import sqlite3
    cur = sqlite3.connect(sys.argv[1]).cursor()
    for name, start, end in cur.execute("select name, start, end from kernels"):
        rows.append((int(start), int(end), clean(name)))
else:
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), clean(r["Kernel_Name"])))
rows.sort()
# split into iterations at select kernels
starts = [i for i, r in enumerate(rows) if "pca_select" in r[2]]
# PCA runs: a gap of more than 5 ms between consecutive selects starts a new run
runs, cur = [], [starts[0]]
for a, b in zip(starts, starts[1:]):
    if rows[b][0] - rows[a][0] > 5_000_000:
        runs.append(cur)
        cur = []
    cur.append(b)
runs.append(cur)
run = runs[-1]
print(f"{len(runs)} PCA runs; last has {len(run)} iterations")
tot_span = tot_busy = 0
agg = defaultdict(float)
for n, (a, b) in enumerate(zip(run, run[1:])):
    span = (rows[b][0] - rows[a][0]) / 1e3
    busy = sum(r[1] - r[0] for r in rows[a:b]) / 1e3
    per = defaultdict(float)
    for r in rows[a:b]:
        per[r[2]] += (r[1] - r[0]) / 1e3
    if n >= 12:
        tot_span += span
        tot_busy += busy
        for k, v in per.items():
            agg[k] += v
    if n < 14 or n % 8 == 0:
        top = ", ".join(f"{k.replace('_kernel','')} {v:.0f}" for k, v in sorted(per.items(), key=lambda x: -x[1])[:6])
        print(f"iter {n:3d} span {span:7.1f} us busy {busy:7.1f} us  kernels {b-a:3d} | {top}")
print(f"tail (iter >= 12): span {tot_span/1e3:.2f} ms, busy {tot_busy/1e3:.2f} ms")
for k, v in sorted(agg.items(), key=lambda x: -x[1]):
    print(f"   {k:32s} {v/1e3:6.2f} ms")
