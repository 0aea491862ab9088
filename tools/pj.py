"""Print the headline fields of a bench.py JSON line: python tools/pj.py file.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["kernels_ms_per_step"])
if d.get("check"):
    print([(g["correl"], g["argmax_mismatch"], g["ok"]) for g in d["check"]["glr"]], d["check"]["ok"])
