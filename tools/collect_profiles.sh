#!/bin/bash
# Collect the end-of-round profile set into gpurun_out/<tag>/ (run on the GPU box from the repo root): tools/collect_profiles.sh <tag>
TAG=${1:-r02b}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. serialised steady-state step: per-kernel totals (20 + 3 steps, nothing else in the trace)
OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 STEPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -o step -- python $R/tools/step_profile.py > $O/step.log 2>&1
# 2. counters (own pass, no trace domains): L2 <-> memory requests and hit rate per kernel symbol
OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 STEPS=3 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc -o pmc -- python $R/tools/step_profile.py > $O/pmc.log 2>&1
# 3. the bench command under the kernel trace, and unprofiled
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
# 4. the HBM kernels at the decoder shape (rocprof rows for bench.py's hbm_kernels block)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hbm -o hbm -- python $R/tools/lndw_probe.py > $O/hbm_probe.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc $O/pmc_glds "OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 STEPS=3 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -- python tools/step_profile.py" > /dev/null 2>&1
python bench.py > $O/bench_default.log 2>&1
OSP_DISC_STREAMS=0 OSP_VOC_STREAM=0 TOP=400 python tools/gemm_inventory.py > $O/gemm_inventory.txt 2>&1
python tools/gemm_small_probe.py > $O/gemm_small_probe.txt 2>&1
OSP_GEMM_SMALL=0 TAG=reg python tools/gemm_small_probe.py >> $O/gemm_small_probe.txt 2>&1
python tools/lstm_probe.py > $O/lstm_probe.txt 2>&1
python tools/lndw_probe.py > $O/lndw_probe.txt 2>&1
rm -f $O/*/*_kernel_trace.csv $O/pmc/*counter_collection.csv $O/*/*.db
python tools/stats_per_step.py $O/step/step_kernel_stats.csv 23 12
tail -1 $O/bench_default.log | cut -c1-300
