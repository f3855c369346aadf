#!/bin/bash
# tools/probes/chain.sh for the Transformer backbone (configs[3]): generator chain of one serialised step, kernel by kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05f
rm -rf /tmp/ch && PIPELINE=0 OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 OSP_WGRAD_STREAM=0 OSP_TAPES=0 STEPS=6 rocprofv3 --kernel-trace --output-format csv -d /tmp/ch -- python $R/tools/step_profile_tf.py > /tmp/ch.log 2>&1
f=$(find /tmp/ch -name "*kernel_trace.csv" | head -1)
python $R/tools/probes/chain_list.py $f > $R/gpurun_out/r05f/chain_tf.txt 2>&1
