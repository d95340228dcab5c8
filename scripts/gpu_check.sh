#!/bin/bash
# Iteration loop: GPU parity tests, block-order timing, bench rates.
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 200 python scripts/gpu_order_timing.py 2>&1 | grep enc || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | grep -o '"value[^,]*,\|compress_GBps_per_gpu[^,]*,[^,]*'
