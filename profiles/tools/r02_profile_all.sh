cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_pmc
mkdir -p $O
python3 $R/profiles/tools/gemm_one.py 20 > $O/gemm_one_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gemm_stats -- python3 $R/profiles/tools/gemm_one.py 20 > $O/gemm_stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/gemm_sq -- python3 $R/profiles/tools/gemm_one.py 5 > $O/gemm_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/gemm_fetch -- python3 $R/profiles/tools/gemm_one.py 5 > $O/gemm_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/gemm_write -- python3 $R/profiles/tools/gemm_one.py 5 > $O/gemm_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/gemm_tcc -- python3 $R/profiles/tools/gemm_one.py 5 > $O/gemm_tcc.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/attn_sq -- python3 $R/profiles/tools/attn_one.py 3 > $O/attn_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/attn_lds -- python3 $R/profiles/tools/attn_one.py 3 > $O/attn_lds.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/attn_stats -- python3 $R/profiles/tools/attn_one.py 10 > $O/attn_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_stats.log 2>&1
ls $O/*/*/ | head -40
