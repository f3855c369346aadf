#!/bin/bash
O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests/test_tape.py tests/test_gpu_tape.py -x -q > $O/test_tape.log 2>&1; echo "tape tests rc $?" | tee $O/rc.txt
python bench.py --no-cpu-baseline --no-infer --no-am-only > $O/bench_tapes.log 2>&1; tail -1 $O/bench_tapes.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tapes on ', d['ms_per_step'], d['host_enqueue_ms_per_step'], d['call_tapes'])" | tee -a $O/rc.txt
python tools/host_only_probe.py > $O/host_only_tapes.log 2>&1
python tools/cpu_profile.py > $O/cpu_profile_tapes.txt 2>&1
python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "gpu suite rc $?" | tee -a $O/rc.txt
tail -25 $O/test_tape.log; grep -i "warn" $O/bench_tapes.log | head; tail -4 $O/host_only_tapes.log; tail -15 $O/gputest.log
