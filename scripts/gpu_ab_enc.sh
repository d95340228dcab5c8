#!/bin/bash
# A/B of encoder variant libraries: LZ-class launch and the bench / 8 GiB encode launches.
for v in "$@"; do
  lib=$PWD/lz4_amd/liblz4mi_$v.so
  [ "$v" = base ] && lib=$PWD/lz4_amd/liblz4mi.so
  echo "== $v"
  LZ4MI_LIB=$lib timeout -k 10 200 python scripts/gpu_scale_timing.py 6 2048 4096 2>&1 | grep class || exit 1
  for extra in "" "--blocks 131072 --steps 4 --warmup 1"; do
    LZ4MI_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline $extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k: round(v['avg_launch_ms'],3) for k,v in d['roofline_all'].items()})" || exit 1
  done
done
