#!/bin/bash
# Decoder iteration: parity tests that touch the decoder, per-class timing, the bench line.
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_formats.py -m gpu -x -q -k "decoder or golden or frame or decode or stream or linked or full_size" --timeout 120 --timeout-method thread > gpurun_out/dec_tests.log 2>&1 || { tail -30 gpurun_out/dec_tests.log; exit 1; }
tail -2 gpurun_out/dec_tests.log
timeout -k 10 200 python scripts/gpu_class_timing.py 2>&1 | grep class | tee gpurun_out/class_timing_iter.log || exit 1
timeout -k 10 200 python bench.py > gpurun_out/bench_iter.json 2> gpurun_out/bench_iter.err || { tail gpurun_out/bench_iter.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/bench_iter.json'));print(d['value'], {k:(round(v['avg_launch_ms'],3),round(v['frac'],3)) for k,v in d['roofline_all'].items()})"
