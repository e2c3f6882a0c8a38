#!/bin/bash
# rocprofv3 kernel trace of a short bench run + per-queue summary of one step (run from the repo root on the GPU box)
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-table > /tmp/prof_tr.log 2>&1
f=$(find /tmp/prof_tr -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/tools/trace_steps.py $f --small --gaps
python3 $R/profiles/tools/trace_steps.py $f --timeline > $R/gpurun_out/tr_timeline.txt
g=$(find /tmp/prof_tr -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out/tr && cp $g $R/gpurun_out/tr/kernel_stats.csv
cp $f $R/gpurun_out/tr/kernel_trace.csv
