#!/bin/bash
# rocprofv3 kernel trace of the free-running model-API (drop-in) loop, 12 steps at B = 64: per-kernel totals (run from the repo root on the GPU box)
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_di
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_di -- python3 $R/profiles/tools/dropin_phases.py ${1:-64} 10 free > /tmp/prof_di.log 2>&1
tail -n 6 /tmp/prof_di.log
g=$(find /tmp/prof_di -name "*kernel_stats.csv" | head -1)
f=$(find /tmp/prof_di -name "*kernel_trace.csv" | head -1)
mkdir -p $R/gpurun_out/di && cp $g $R/gpurun_out/di/kernel_stats.csv && cp $f $R/gpurun_out/di/kernel_trace.csv
python3 - "$g" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"all kernels: {tot / 1e6:.1f} ms over 12 steps = {tot / 12e6:.2f} ms of kernel time per step")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 12e6:8.3f} ms/step {float(r['AverageNs']) / 1e3:8.1f} us")
P
