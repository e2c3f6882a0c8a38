#!/bin/bash
# Per-phase split of a training step from the engine's roctx ranges (MV_ROCTX=2: a range per phase, the device synchronised at every phase
# boundary so that a range lasts as long as its kernels -- the two-stream overlap is given up for the measurement).
# rocprofv3 --marker-trace of a short bench run; the ranges are summed per phase name by phase_ranges.py.  Run from the repo root on the GPU box.
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_mk
export MV_ROCTX=2
rocprofv3 --marker-trace --output-format csv -d /tmp/prof_mk -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-table > /tmp/prof_mk.log 2>&1
f=$(find /tmp/prof_mk -name "*marker*trace*.csv" | head -1)
echo "marker trace: $f"
python3 $R/profiles/tools/phase_ranges.py "$f"
