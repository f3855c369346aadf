#!/bin/bash
# SQ / LDS counters of the 64 <- 64 layer-1 forward (ONLY=1: one shape of tools/probes/panel_probe.py), panel kernel vs per-tap kernel
TAG=${1:-r06pmc}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_LDS[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*\|SQ_ACTIVE_INST[A-Z_0-9]*\|TCP_[A-Z_0-9]*" $O/avail.txt | sort -u > $O/avail_short.txt
S1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"
S2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
S3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
for V in 1 0; do
  for S in 1 2 3; do
    eval "SET=\$S$S"
    OSP_N64_PANEL=$V ONLY=1 timeout 600 rocprofv3 --pmc $SET --output-format csv -d $O/p${V}_$S -o pmc -- python $R/tools/probes/panel_probe.py > $O/p${V}_$S.log 2>&1
  done
  python $R/tools/pmc_table.py $O/p${V}_1,$O/p${V}_2,$O/p${V}_3 n64 > $O/panel${V}_counters.txt 2>&1
done
rm -rf $O/p?_?
cd $R
