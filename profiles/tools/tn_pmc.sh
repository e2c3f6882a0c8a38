#!/bin/bash
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for m in tn nt; do
python3 $R/profiles/tools/gemm_tn_one.py 10 25483 $m
rm -rf /tmp/p1 /tmp/p2
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/p1 -- python3 $R/profiles/tools/gemm_tn_one.py 3 25483 $m > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/p2 -- python3 $R/profiles/tools/gemm_tn_one.py 3 25483 $m > /dev/null 2>&1
python3 $R/profiles/tools/pmc_summary.py $(find /tmp/p1 /tmp/p2 -name "*counter_collection.csv") --match gemm_
done
