R=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_dm
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dm -- python3 $R/profiles/tools/dominant.py attn 100 20 > /tmp/prof_dm.log 2>&1
g=$(find /tmp/prof_dm -name "*kernel_stats.csv" | head -1)
python3 - "$g" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "attn" in r["Name"]: print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3)
P
