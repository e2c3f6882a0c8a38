#!/bin/bash
# Round-5 profile passes (run from the repo root on the GPU box).  For EVERY MFMA kernel family of the step (profiles/tools/dominant.py:
# ffn1, dw, attn and the ten other GEMM calls of a layer), in ONE lease on ONE box:
#   1. the un-profiled HIP-event timing bench.py uses (50 warm-up + 200 timed launches, median / min / max of 10-launch batches),
#   2. rocprofv3 --kernel-trace --stats of the SAME command (per-launch durations -> average over all launches, average and median after the
#      warm-up launches),
#   3. rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (HBM bytes per launch, gfx950 corrections per MI355X_MICROARCH.md);
#      for attention also the SQ and TCC groups, for both block orders (MV_ATTN_ORDER).
# Raw output under gpurun_out/r5_pmc; profiles/tools/r05_condense.py writes the committed summaries (profiles/r05_*).
# usage: profiles/tools/r05_profile_all.sh [cases...]      (default: all 13)
R=$(pwd); cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r5_pmc
mkdir -p $O
CASES="$@"
[ -z "$CASES" ] && CASES="ffn1 dw attn qkv wo ffn2 dz da dctx dxqkv dw2 dwqkv dwo"
for c in $CASES; do
  rm -rf $O/${c}_*
  python3 $R/profiles/tools/dominant.py $c 200 50 > $O/${c}_plain.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_stats -- python3 $R/profiles/tools/dominant.py $c 200 50 > $O/${c}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/${c}_fetch -- python3 $R/profiles/tools/dominant.py $c 10 5 > $O/${c}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${c}_write -- python3 $R/profiles/tools/dominant.py $c 10 5 > $O/${c}_write.log 2>&1
  if [ "$c" = "attn" ] || [ "$c" = "ffn1" ] || [ "$c" = "dw" ]; then
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/${c}_sq -- python3 $R/profiles/tools/dominant.py $c 10 5 > $O/${c}_sq.log 2>&1
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d $O/${c}_tcc -- python3 $R/profiles/tools/dominant.py $c 10 5 > $O/${c}_tcc.log 2>&1
  fi
  echo "done $c $(date +%T)"
done
if [ -z "$1" ] || [ "$MV_BENCH_STATS" = "1" ]; then
  rm -rf $O/bench_stats
  export MV_BENCH_NO_KERNELS=1      # the step's kernels only (bench.py otherwise times the dominant kernels alone, 250 launches each)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-table > $O/bench_stats.log 2>&1
  unset MV_BENCH_NO_KERNELS
fi
python3 $R/profiles/tools/r05_condense.py $O $R/gpurun_out/r5_summary
