#!/bin/bash
# Builds variant libraries of the attention translation unit with -DATT_ABL=<bits> (timing experiments: see mv_attn.hip) next to the
# product objects and times the forward kernel with each.  usage (on the GPU box): profiles/tools/attn_ablate.sh "0 1 2 4 ..."
# The variant libraries are built on the CPU side first:  profiles/tools/attn_ablate.sh build "0 1 2 4"
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
PKG="$ROOT/multi-modality-self-supervision_amd"
VAR="$PKG/build/variants"
if [ "$1" = "build" ]; then
  mkdir -p "$VAR"
  for v in $2; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DATT_ABL=$((v % 100000)) $EXTRA_DEFS -c "$PKG/csrc/mv_attn.hip" -o "$VAR/mv_attn_abl$v.o" &
  done
  wait
  for v in $2; do
    objs=$(ls "$PKG"/build/*.o | grep -v mv_attn.o)
    hipcc --offload-arch=gfx950 -shared -fPIC -o "$VAR/libmedvill_abl$v.so" $objs "$VAR/mv_attn_abl$v.o"
  done
  exit 0
fi
for v in $1; do
  echo -n "ATT_ABL=$v: "
  MV_LIB_PATH="$VAR/libmedvill_abl$v.so" python "$ROOT/profiles/tools/attn_bench.py" f16 full512 2>/dev/null | grep "p=0.1" | head -1
done
