#!/bin/bash
# Decoder iteration loop: decoder parity tests first, then the whole GPU suite, per-class timing, bench rates.
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "decoder or golden or frame" --timeout 120 --timeout-method thread > gpurun_out/dec_tests.log 2>&1 || { tail -30 gpurun_out/dec_tests.log; exit 1; }
tail -2 gpurun_out/dec_tests.log
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 200 python scripts/gpu_class_timing.py 2>&1 | grep class | tee gpurun_out/class_timing_r2.log || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/bench_r2.log 2>&1 || { tail -5 gpurun_out/bench_r2.log; exit 1; }
grep -o '"value[^,]*,\|compress_GBps_per_gpu[^,]*,[^,]*\|"roofline_all.*' gpurun_out/bench_r2.log
