#!/bin/bash
# first GPU run of round 4: tape machinery + taped discriminator stacks vs eager, host-enqueue A/B, MFMA-busy counters, full-size goldens
O=gpurun_out/r04a; mkdir -p $O
python -m pytest tests/test_tape.py tests/test_gpu_tape.py -x -q > $O/test_tape.log 2>&1; echo "tape tests rc $?" | tee $O/rc.txt
python bench.py --no-cpu-baseline --no-infer --no-am-only > $O/bench_tapes.log 2>&1; tail -1 $O/bench_tapes.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tapes on ', d['ms_per_step'], d['host_enqueue_ms_per_step'], d['call_tapes'])" | tee -a $O/rc.txt
OSP_TAPES=0 python bench.py --no-cpu-baseline --no-infer --no-am-only > $O/bench_notapes.log 2>&1; tail -1 $O/bench_notapes.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tapes off', d['ms_per_step'], d['host_enqueue_ms_per_step'])" | tee -a $O/rc.txt
python tools/host_only_probe.py > $O/host_only_tapes.log 2>&1; OSP_TAPES=0 python tools/host_only_probe.py > $O/host_only_notapes.log 2>&1
python tools/cpu_profile.py > $O/cpu_profile_tapes.txt 2>&1
python -m pytest tests/test_gpu_fullsize_golden.py -q > $O/test_fullsize.log 2>&1; echo "fullsize tests rc $?" | tee -a $O/rc.txt
bash tools/pmc_mfma.sh r04a > $O/pmc_mfma_run.log 2>&1
tail -5 $O/test_tape.log; tail -15 $O/test_fullsize.log; tail -3 $O/host_only_tapes.log $O/host_only_notapes.log; head -20 $O/pmc_mfma_busy.txt
