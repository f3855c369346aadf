#!/bin/bash
# TCC counters of the weight-gradient probe (own pass, no trace domains): tools/probes/wgrad_pmc.sh <tag> [env...]
TAG=$1; shift
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" REP=3 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc -o pmc -- python $R/tools/probes/wgrad_shapes.py > $O/pmc.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc $O/pmc_wgrad "wgrad_shapes.py" > /dev/null 2>&1
rm -rf $O/pmc
cat $O/pmc_wgrad.txt | head -30
