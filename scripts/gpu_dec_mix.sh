#!/bin/bash
# Queue mix of the decoder (streamed blocks per heavy block) at 1 GiB and 8 GiB per launch.
for m in 0 1 2 3 4; do
  for nb in 16384 131072; do
    LZ4MI_DEC_MIX=$m timeout -k 10 300 python bench.py --blocks $nb --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mix $m blocks $nb', round(d['value'],1), {k: (round(v['avg_launch_ms'],3), round(v['frac'],3)) for k,v in d['roofline_all'].items()})" || exit 1
  done
done
