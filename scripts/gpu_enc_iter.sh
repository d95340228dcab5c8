#!/bin/bash
# Encoder iteration: parity tests (every block of the full-size corpus among them), the bench line, the 8 GiB line.
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_formats.py tests/test_gpu_frame_memory.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/enc_tests.log 2>&1 || { tail -30 gpurun_out/enc_tests.log; exit 1; }
tail -2 gpurun_out/enc_tests.log
for extra in "" "--blocks 131072"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $extra > gpurun_out/bench_iter.json 2> gpurun_out/bench_iter.err || { tail gpurun_out/bench_iter.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/bench_iter.json'));print(round(d['value'],1), {k:(round(v['avg_launch_ms'],3),round(v['frac'],3)) for k,v in d['roofline_all'].items()})"
done
