#!/bin/bash
# Same-box A/B of the attention kernels of two library builds (stand-alone, profiles/tools/attn_bench.py f16 full512).
# usage: profiles/tools/ab_attn.sh <libA.so> <libB.so> [rounds]
A="$1"; B="$2"; N="${3:-3}"
for i in $(seq 1 $N); do
  for L in "$A" "$B"; do
    echo -n "$(basename $L): "; MV_LIB_PATH="$L" python profiles/tools/attn_bench.py f16 full512 2>/dev/null | grep "p=0.1" | cut -c1-100
  done
done
