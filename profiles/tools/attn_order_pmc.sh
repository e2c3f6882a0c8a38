R=$(pwd); cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r5_pmc_order1
rm -rf $O; mkdir -p $O
export MV_KNOBS=attn_order=1
python3 $R/profiles/tools/dominant.py attn 100 30 > $O/attn_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/attn_stats -- python3 $R/profiles/tools/dominant.py attn 200 50 > $O/attn_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/attn_fetch -- python3 $R/profiles/tools/dominant.py attn 10 5 > $O/attn_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/attn_write -- python3 $R/profiles/tools/dominant.py attn 10 5 > $O/attn_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/attn_sq -- python3 $R/profiles/tools/dominant.py attn 10 5 > $O/attn_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d $O/attn_tcc -- python3 $R/profiles/tools/dominant.py attn 10 5 > $O/attn_tcc.log 2>&1
unset MV_KNOBS
python3 $R/profiles/tools/r05_condense.py $O $R/gpurun_out/r5_summary_order1
