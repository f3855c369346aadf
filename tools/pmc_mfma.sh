#!/bin/bash
# MFMA-busy / SQ counters per kernel symbol of the steady-state training step (VERDICT r03 item 4 i; north_star: "MFMA-busy counters
# reported against gfx950 peak").  Own pass, counters only (no trace domains -- gpurun refuses the combination).
#   tools/pmc_mfma.sh <tag>       -> gpurun_out/<tag>/pmc_mfma/ , gpurun_out/<tag>/pmc_mfma_busy.{json,txt}
TAG=${1:-r04}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1
FULL="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
SAFE="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
for SET in "$FULL" "$SAFE"; do
  rm -rf $O/pmc_mfma
  OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 OSP_TAPES=${OSP_TAPES:-1} STEPS=2 timeout 900 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_mfma -o pmc -- python $R/tools/step_profile.py > $O/pmc_mfma.log 2>&1
  if ls $O/pmc_mfma/*counter_collection.csv $O/pmc_mfma/*/*counter_collection.csv >/dev/null 2>&1; then echo "counters: $SET" > $O/pmc_mfma_set.txt; break; fi
done
cd $R
python tools/pmc_mfma_summary.py $O/pmc_mfma $O/pmc_mfma_busy "OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 STEPS=2 rocprofv3 --pmc $(cat $O/pmc_mfma_set.txt | cut -d: -f2) --output-format csv -- python tools/step_profile.py"
rm -f $O/pmc_mfma/*counter_collection.csv $O/pmc_mfma/*/*counter_collection.csv $O/pmc_mfma/*.db $O/pmc_mfma/*/*.db
