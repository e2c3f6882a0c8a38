#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel from one or more counter_collection.csv files.
usage: pmc_summary.py <csv> [<csv> ...] [--match substring]"""
import collections
import csv
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
if match in args:
    args.remove(match)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in args:
    with open(fn) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if match and match not in k:
                continue
            acc[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}   (n={len(v)})")
