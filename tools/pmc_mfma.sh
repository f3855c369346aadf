#!/bin/bash
# MFMA-busy / SQ counters per kernel symbol of the steady-state training step (VERDICT r03 item 4 i; north_star: "MFMA-busy counters
# reported against gfx950 peak").  Own passes, counters only (no trace domains -- gpurun refuses the combination); two passes
# because the SQ block has 8 slots:  A = matrix-pipe occupancy, B = where the waves' time goes.
#   tools/pmc_mfma.sh <tag>       -> gpurun_out/<tag>/pmc_mfma_busy.{json,txt}
TAG=${1:-r04}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
B="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
rm -rf $O/pmc_mfma_a $O/pmc_mfma_b
OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 OSP_TAPES=${OSP_TAPES:-1} STEPS=2 timeout 900 rocprofv3 --pmc $A --output-format csv -d $O/pmc_mfma_a -o pmc -- python $R/tools/step_profile.py > $O/pmc_mfma_a.log 2>&1
OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 OSP_TAPES=${OSP_TAPES:-1} STEPS=2 timeout 900 rocprofv3 --pmc $B --output-format csv -d $O/pmc_mfma_b -o pmc -- python $R/tools/step_profile.py > $O/pmc_mfma_b.log 2>&1
cd $R
python tools/pmc_mfma_summary.py $O/pmc_mfma_a,$O/pmc_mfma_b $O/pmc_mfma_busy "OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 STEPS=2 rocprofv3 --pmc <set> --output-format csv -- python tools/step_profile.py; set A = $A; set B = $B"
rm -rf $O/pmc_mfma_a $O/pmc_mfma_b
