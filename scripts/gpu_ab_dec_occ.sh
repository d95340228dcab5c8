#!/bin/bash
# A/B of decoder residency variants (scripts/build_variant.sh <name> "-DLZ4MI_DEC_WPE=.. -DLZ4MI_HIST=.."):
# per-class launches, the bench launch, configs[4]'s decode and the 8 GiB launch, for each library named.
mkdir -p gpurun_out/r3
for v in "$@"; do
  lib=$PWD/lz4_amd/liblz4mi_$v.so
  [ "$v" = base ] && lib=$PWD/lz4_amd/liblz4mi.so
  echo "== $v"
  LZ4MI_LIB=$lib timeout -k 10 200 python scripts/gpu_class_timing.py 2>&1 | grep "class [0356]" || exit 1
  LZ4MI_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-by-class 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1gib', round(d['value'],1), {k: round(v['avg_launch_ms'],3) for k,v in d['roofline_all'].items()})" || exit 1
  LZ4MI_LIB=$lib timeout -k 10 400 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config4', d['phase_ms'])" || exit 1
  LZ4MI_LIB=$lib timeout -k 10 400 python bench.py --blocks 131072 --steps 3 --warmup 1 --no-cpu-baseline --no-by-class 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8gib', round(d['value'],1), {k: round(v['avg_launch_ms'],3) for k,v in d['roofline_all'].items()})" || exit 1
done
