#!/bin/bash
# Encoder iteration loop: parity tests, one LZ class at two batch sizes, bench compress rate.
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 120 python scripts/gpu_scale_timing.py 5 64 4096 2>&1 | grep class || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | grep -o "compress_GBps_per_gpu[^,]*,[^,]*"
