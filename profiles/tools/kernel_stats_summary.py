#!/usr/bin/env python3
"""Condensed table of a rocprofv3 --kernel-trace --stats kernel_stats.csv.  usage: kernel_stats_summary.py <kernel_stats.csv> [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print(f"{'kernel':88s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}   min / max us")
for r in rows[:top]:
    n = r["Name"].replace("void ", "").replace("(GemmArgs)", "").replace("(AttnArgs)", "").replace("(GemmArgs, int, int)", "")
    print(f"{n[:88]:88s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:9.2f} {float(r['AverageNs']) / 1e3:9.1f} "
          f"{float(r['Percentage']):6.2f}   {float(r['MinNs']) / 1e3:7.1f} / {float(r['MaxNs']) / 1e3:7.1f}")
print(f"(all kernels: {sum(float(r['TotalDurationNs']) for r in rows) / 1e6:.1f} ms of kernel time)")
