#!/bin/bash
# Quick loop: all GPU tests, then the default bench line twice (value, per-kernel launch ms and roofline fraction).
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gputest.log 2>&1 || { tail -30 gpurun_out/gputest.log; exit 1; }
tail -1 gpurun_out/gputest.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k:(round(v['avg_launch_ms'],4), round(v['frac'],4)) for k,v in d['roofline_all'].items()})" || exit 1
done
