#!/usr/bin/env python3
"""Sum the roctx ranges of a rocprofv3 --marker-trace CSV per phase name (layers folded together).  usage: phase_ranges.py <marker_api_trace.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
if not rows:
    sys.exit("no marker records")
cols = rows[0].keys()
name_col = next((c for c in ("Function", "Name", "Message", "Marker_Name") if c in cols), None)
s_col = next((c for c in ("Start_Timestamp", "Begin_Timestamp", "Start") if c in cols), None)
e_col = next((c for c in ("End_Timestamp", "End") if c in cols), None)
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = re.sub(r"layer\d+", "layer*", r[name_col])
    acc[n][0] += 1
    acc[n][1] += (int(r[e_col]) - int(r[s_col])) / 1e6
tot = sum(v[1] for v in acc.values())
steps = max(1, acc.get("embed", [1])[0])
print(f"columns: {list(cols)}")
print(f"{'phase':28s} {'ranges':>7s} {'ms total':>10s} {'ms per step':>12s} {'share':>7s}   ({steps} steps, device synchronised at every boundary)")
for n, (c, t) in sorted(acc.items(), key=lambda x: -x[1][1]):
    print(f"{n:28s} {c:7d} {t:10.2f} {t / steps:12.3f} {100 * t / tot:6.1f} %")
