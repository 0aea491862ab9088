#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters from its rocpd sqlite output (ROCm 7.2 default).

    tools/pmc_db.py OUT.json DIR [DIR ...]      (each DIR: the -d directory of one --pmc pass)

Values are summed over XCDs / SEs as rocprofv3 reports them and divided by the kernel's launches.
SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_BUSY_CYCLES is summed
over the 32 shader engines; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md)."""
import collections
import glob
import json
import re
import sqlite3
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    res = collections.defaultdict(dict)
    for d in dirs:
        for f in glob.glob(d + "/**/*_results.db", recursive=True):
            cur = sqlite3.connect(f).cursor()
            rows = cur.execute(
                "select kernel_name, counter_name, sum(value), count(distinct dispatch_id), "
                "avg(duration) from counters_collection group by kernel_name, counter_name")
            for k, c, v, n, dur in rows.fetchall():
                m = re.search(r"([a-z_0-9]+_kernel)", k)
                k = m.group(1) if m else k[:48]
                res[k][c] = v / n
                res[k]["launches"] = n
                res[k]["avg_duration_us"] = round(dur / 1e3, 1)
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("avg_duration_us", 0)):
        print(k)
        for c, x in sorted(v.items()):
            print(f"    {c:28s} {x:18.1f}")


if __name__ == "__main__":
    main()
