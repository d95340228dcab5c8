#!/bin/bash
# Decoder iteration loop: decoder parity tests, phase breakdown (diagnostic build), per-class timing.
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_formats.py -m gpu -x -q -k "decoder or golden or frame or decode or stream or linked" --timeout 120 --timeout-method thread > gpurun_out/dec_tests.log 2>&1 || { tail -30 gpurun_out/dec_tests.log; exit 1; }
tail -2 gpurun_out/dec_tests.log
LZ4MI_LIB=lz4_amd/liblz4mi_prof.so NB=1024 timeout -k 10 200 python scripts/gpu_phase_dec.py 6 > gpurun_out/phase_dec_r2.log 2>&1 || { tail gpurun_out/phase_dec_r2.log; exit 1; }
cat gpurun_out/phase_dec_r2.log
timeout -k 10 200 python scripts/gpu_class_timing.py 2>&1 | grep class | tee gpurun_out/class_timing_r2.log || exit 1
