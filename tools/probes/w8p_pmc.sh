#!/bin/bash
# SQ / LDS counters of the 8-wave conv-GEMM variants on one full-chip shape (256 tiles): own --pmc passes, counters only.
#   tools/probes/w8p_pmc.sh <tag> "<variants>"   -> gpurun_out/<tag>/w8p_pmc_v<variant>.txt
TAG=${1:-w8p}; VARS=${2:-"0 2"}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
B="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
C="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"
for v in $VARS; do
  for s in A B C; do
    rm -rf $O/pmc_$s
    ONE=1 OSP_GEMM_W8P=$v timeout 300 rocprofv3 --pmc ${!s} --output-format csv -d $O/pmc_$s -o pmc -- python $R/tools/probes/w8p_probe.py > $O/pmc_$s.log 2>&1
  done
  (cd $R && python tools/pmc_mfma_summary.py $O/pmc_A,$O/pmc_B,$O/pmc_C $O/w8p_pmc_v$v "OSP_GEMM_W8P=$v rocprofv3 --pmc <set> -- python tools/probes/w8p_probe.py; A = $A; B = $B; C = $C")
  rm -rf $O/pmc_A $O/pmc_B $O/pmc_C
done
