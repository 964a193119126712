"""Aggregate rocprofv3 --pmc CSV output: mean counter value per dispatch for kernels whose name contains a pattern.
usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> <kernel substring> [...]"""
import collections
import csv
import glob
import json
import sys

d, pats = sys.argv[1], sys.argv[2:]
acc = {p: collections.defaultdict(list) for p in pats}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name") or row.get("kernel_name") or ""
            for p in pats:
                if p in name:
                    acc[p][(row["Counter_Name"], row.get("Dispatch_Id") or row.get("dispatch_id"))].append(float(row["Counter_Value"]))
out = {}
for p, m in acc.items():
    per = collections.defaultdict(list)
    for (cn, _disp), vals in m.items():
        per[cn].append(sum(vals))                      # sum over dimensions (XCDs / SEs) of one dispatch
    out[p] = {cn: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for cn, v in sorted(per.items())}
print(json.dumps(out, indent=1))
