#!/bin/bash
# A/B of decoder variants on launches of literal-only blocks and on the bench decode launch.
for v in "$@"; do
  lib=$PWD/lz4_amd/liblz4mi_$v.so
  [ "$v" = base ] && lib=$PWD/lz4_amd/liblz4mi.so
  echo "== $v"
  LZ4MI_LIB=$lib timeout -k 10 200 python scripts/gpu_scale_timing.py 0 16384 2>&1 | grep class || exit 1
  LZ4MI_LIB=$lib timeout -k 10 200 python scripts/gpu_scale_timing.py 3 16384 2>&1 | grep class || exit 1
  LZ4MI_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k: round(v['avg_launch_ms'],3) for k,v in d['roofline_all'].items()})" || exit 1
done
