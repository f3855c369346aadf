#!/bin/bash
O=gpurun_out/r04q; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "gpu suite rc $?" | tee $O/rc.txt
tail -5 $O/gputest.log
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-600
