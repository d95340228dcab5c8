#!/bin/bash
# The other bench lines in short form: one-rank rehearsal of the sharded step, the 8 GiB shard, its rehearsal.
p='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), {k:(round(v["avg_launch_ms"],4), round(v["frac"],4)) for k,v in d["roofline_all"].items()})'
LZ4MI_BENCH_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "$p" || exit 1
timeout -k 10 500 python bench.py --blocks 131072 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$p" || exit 1
LZ4MI_BENCH_DIST=1 timeout -k 10 500 python bench.py --corpus-gib 8 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$p" || exit 1
