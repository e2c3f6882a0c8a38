#!/bin/bash
# SURVEY 5.2 / VERDICT r3 item 10: the HOST side of the C-ABI library under AddressSanitizer + UBSan, CPU only (GPU ASan / xnack builds
# are not available on the pool and are never attempted).  hipcc compiles every translation unit twice (host x86-64, device gfx950);
# `-Xarch_host` instruments the host half only: the launchers' argument validation, the descriptor / workspace checks and the one
# host function (mv_mask_verify_host).  The instrumented library goes to a scratch directory (never into the package), is selected
# through MV_LIB_PATH, and the CPU-reachable ABI tests run against it with the ASan runtime preloaded into python.
#   usage: tools/asan_host_check.sh            (exit code 0 = no sanitizer report, all tests passed)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/multi-modality-self-supervision_amd/csrc"
OUT="${MV_ASAN_DIR:-/tmp/medvill_asan}"
HIPCC="${HIPCC:-$(command -v hipcc || echo /opt/rocm/bin/hipcc)}"
ASAN_RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
mkdir -p "$OUT"
SAN=(-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -fno-sanitize-recover=undefined)
pids=()
for f in "$SRC"/*.hip; do
  o="$OUT/$(basename "${f%.hip}").o"
  if [ ! -e "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find "$SRC" "$ROOT/include" -name '*.h' -newer "$o" | head -1)" ]; then
    "$HIPCC" --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-value "${SAN[@]}" -c "$f" -o "$o" &
    pids+=($!)
    if [ "${#pids[@]}" -ge 4 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
# the test / experiment library: the same objects with the debug build of mv_api.hip (include/medvill_debug.h); kept out of "$OUT"/*.o
mkdir -p "$OUT/dbg"
"$HIPCC" --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-value -DMV_DEBUG_KNOBS "${SAN[@]}" -c "$SRC/mv_api.hip" -o "$OUT/dbg/mv_api_dbg.o"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -Xarch_host -fsanitize=address,undefined -o "$OUT/libmedvill_hip_asan.so" "$OUT"/*.o
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -Xarch_host -fsanitize=address,undefined -o "$OUT/libmedvill_hip_asan_dbg.so" $(ls "$OUT"/*.o | grep -v '/mv_api.o$') "$OUT/dbg/mv_api_dbg.o"
cd "$ROOT"
# detect_leaks=0: CPython itself never frees its arenas; halt_on_error: the first report fails the run
env LD_PRELOAD="$ASAN_RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    MV_LIB_PATH="$OUT/libmedvill_hip_asan.so" \
    python -m pytest tests/test_abi.py tests/test_abi_rejects.py -x -q -p no:cacheprovider "$@"
echo "asan_host_check: clean ($(basename "$ASAN_RT"), $(ls "$OUT"/*.o | wc -l) translation units)"
