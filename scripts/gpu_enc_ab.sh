#!/bin/bash
# Encoder A/B on one box: for every library given (paths relative to the repo, default: the release build),
# byte parity of the full-size corpus against the oracle, the 1 GiB launch time, one LZ class alone and loaded.
# usage: bash scripts/gpu_enc_ab.sh [lib ...]
mkdir -p gpurun_out/r3
libs=${@:-lz4_amd/liblz4mi.so}
for lib in $libs; do
  export LZ4MI_LIB=$PWD/$lib
  echo "== $lib"
  timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "full_size_corpus or encoder or fast or frame_encode or shard" 2>&1 | tail -2 || exit 1
  timeout -k 10 120 python scripts/gpu_enc_only.py 2>&1 | grep encode || exit 1
  timeout -k 10 120 python scripts/gpu_scale_timing.py 6 256 4096 2>&1 | grep class || exit 1
done
