#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel over one or more counter_collection.csv files.

    tools/pmc_summary.py OUT.json DIR [DIR ...]

Every DIR is the -d directory of one `rocprofv3 --pmc ... --kernel-trace` pass (separate passes:
SQ has 8 slots, FETCH_SIZE / WRITE_SIZE cannot share one).  Values are summed over the XCDs /
dispatches as rocprofv3 reports them and divided by the number of launches of the kernel.
SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_BUSY_CYCLES counts per
SE; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md, cycle constants)."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"([a-z_0-9]+_kernel)", name)
    return m.group(1) if m else name[:48]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    res = collections.defaultdict(dict)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            agg = collections.defaultdict(lambda: [0.0, set()])
            for r in csv.DictReader(open(f)):
                key = (short(r["Kernel_Name"]), r["Counter_Name"])
                agg[key][0] += float(r["Counter_Value"])
                agg[key][1].add(r["Dispatch_Id"])
            for (k, c), (v, disp) in agg.items():
                res[k][c] = v / max(1, len(disp))
                res[k]["launches"] = len(disp)
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    top = sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("FETCH_SIZE", 0)))[:8]
    for k, v in top:
        keys = ("launches", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES",
                "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT")
        print(f"{k:28s} " + " ".join(f"{c.replace('SQ_', '')}={v[c]:.3g}" for c in keys if c in v))


if __name__ == "__main__":
    main()
