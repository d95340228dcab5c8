#!/bin/bash
# Timing-only builds (wrong output on purpose): what a phase of the encoder costs.  usage: bash scripts/gpu_enc_timing_only.sh lib...
for lib in "$@"; do
  export LZ4MI_LIB=$PWD/$lib
  echo "== $lib"
  timeout -k 10 120 python scripts/gpu_enc_only.py 2>&1 | grep encode
  timeout -k 10 120 python scripts/gpu_scale_timing.py 6 256 4096 2>&1 | grep class | sed 's/dec.*//'
done
