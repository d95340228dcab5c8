#!/bin/bash
# Encoder occupancy experiment: ring size and wavefronts per SIMD (variant libraries), 1 GiB and 8 GiB launches.
run() { LZ4MI_LIB=$PWD/lz4_amd/liblz4mi$1.so LZ4MI_ENC_WAVES_PER_CU=$2 timeout -k 10 300 python bench.py --blocks $3 --steps 4 --warmup 1 --no-cpu-baseline --no-by-class 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 wpc $2 blocks $3', round(d['value'],1), {k: round(v['avg_launch_ms'],3) for k,v in d['roofline_all'].items()})" || exit 1; }
for nb in 16384 131072; do
  run "" 16 $nb; run _w4k 16 $nb; run _w5 20 $nb; run _w6 24 $nb
done
