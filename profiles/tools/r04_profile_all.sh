#!/bin/bash
# Round-4 profile passes (run from the repo root on the GPU box): per-kernel stats and PMC counters (separate passes, as the MI355X
# guide prescribes) of the dominant kernels on the operands bench.py times them on (profiles/tools/dominant.py), and the kernel
# statistics of a short bench run.  Raw output under gpurun_out/r4_pmc; condensed into profiles/r04_* by r04_condense.py.
R=$(pwd); cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r4_pmc
rm -rf $O; mkdir -p $O
for c in dw ffn1 attn; do
  python3 $R/profiles/tools/dominant.py $c 20 > $O/${c}_plain.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_stats -- python3 $R/profiles/tools/dominant.py $c 20 > $O/${c}_stats.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/${c}_sq -- python3 $R/profiles/tools/dominant.py $c 5 > $O/${c}_sq.log 2>&1
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/${c}_fetch -- python3 $R/profiles/tools/dominant.py $c 5 > $O/${c}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${c}_write -- python3 $R/profiles/tools/dominant.py $c 5 > $O/${c}_write.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d $O/${c}_tcc -- python3 $R/profiles/tools/dominant.py $c 5 > $O/${c}_tcc.log 2>&1
  echo "done $c"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-table > $O/bench_stats.log 2>&1
python3 $R/profiles/tools/r04_condense.py $O $R/gpurun_out/r4_summary
