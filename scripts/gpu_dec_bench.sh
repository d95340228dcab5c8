#!/bin/bash
# Decoder check + bench rates: parity tests, debug corpus check, per-class timing, bench.
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_formats.py -m gpu -x -q -k "decoder or golden or frame or decode or stream or linked" --timeout 120 --timeout-method thread > gpurun_out/dec_tests.log 2>&1 || { tail -30 gpurun_out/dec_tests.log; exit 1; }
tail -1 gpurun_out/dec_tests.log
timeout -k 5 120 python scripts/gpu_dec_debug.py 2>&1 | grep -c None
timeout -k 10 200 python scripts/gpu_class_timing.py 2>&1 | grep "class [3456]"
timeout -k 10 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline_all'])"
