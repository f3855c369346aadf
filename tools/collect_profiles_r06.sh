#!/bin/bash
# Round-6 profile set (run on the GPU box from the repo root): tools/collect_profiles_r06.sh [tag]  -> gpurun_out/<tag>/
TAG=${1:-r06}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. serialised steady-state step (streams off): per-kernel totals over 23 steps + launches by (kernel, grid)
OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 STEPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -o step -- python $R/tools/step_profile.py > $O/step.log 2>&1
python $R/tools/trace_shapes.py $O/step/step_kernel_trace.csv 300 > $O/step_kernel_shapes.txt 2>&1
# 2. TCC counters (own pass)
OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 STEPS=3 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc -o pmc -- python $R/tools/step_profile.py > $O/pmc.log 2>&1
# 3. the bench command under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-transformer --extras $O/bench_extras_under_rocprof.json > $O/bench_under_rocprof.log 2>&1
# 4. synthesise()
REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $O/synth -o synth -- python $R/tools/synth_profile.py > $O/synth.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc $O/pmc_glds "OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 STEPS=3 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -- python tools/step_profile.py" > /dev/null 2>&1
# 5. SQ / GRBM counters: MFMA-busy per symbol (two passes)
bash tools/pmc_mfma.sh $TAG > $O/pmc_mfma_run.log 2>&1
rm -f $O/*/*_kernel_trace.csv $O/pmc/*counter_collection.csv $O/*/*.db
python tools/stats_per_step.py $O/step/step_kernel_stats.csv 23 12
head -8 $O/pmc_mfma_busy.txt
