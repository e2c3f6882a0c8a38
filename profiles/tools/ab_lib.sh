#!/bin/bash
# Same-box A/B of two builds of the library on the whole training step: bench.py (headline only) alternately with MV_LIB_PATH=<A> and <B>.
# usage: profiles/tools/ab_lib.sh <libA.so> <libB.so> [rounds]
A="$1"; B="$2"; N="${3:-3}"
for i in $(seq 1 $N); do
  for L in "$A" "$B"; do
    MV_LIB_PATH="$L" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $L)', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'pairs/s')"
  done
done
