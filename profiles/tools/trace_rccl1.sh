#!/bin/bash
# kernel trace of the step with the RCCL path active on a one-rank group (MV_DP_FORCE=1): which RCCL kernels run, for how long
R=$(pwd); cd /tmp && export TMPDIR=/tmp
export MV_DP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
rm -rf /tmp/prof_rc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rc -- python3 $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /tmp/prof_rc.log 2>&1
g=$(find /tmp/prof_rc -name "*kernel_stats.csv" | head -1)
python3 $R/profiles/tools/kernel_stats_summary.py $g 14
f=$(find /tmp/prof_rc -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/tools/trace_steps.py $f > /tmp/prof_rc_steps.txt 2>&1; head -14 /tmp/prof_rc_steps.txt
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_rc.log
mkdir -p $R/gpurun_out/tr_rccl && cp $f $R/gpurun_out/tr_rccl/kernel_trace.csv
