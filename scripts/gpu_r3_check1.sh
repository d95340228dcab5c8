#!/bin/bash
# Round-3 first check: GPU test suite, host-path content checksum before/after, config 2 line.
set -e
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3/pytest.log 2>&1 || { tail -30 gpurun_out/r3/pytest.log; exit 1; }
tail -3 gpurun_out/r3/pytest.log
LZ4MI_LIB=$PWD/lz4_amd/liblz4mi_r2.so timeout -k 10 120 python scripts/gpu_host_path_timing.py ck > gpurun_out/r3/host_path_r2.log 2>&1 && cat gpurun_out/r3/host_path_r2.log
timeout -k 10 120 python scripts/gpu_host_path_timing.py ck > gpurun_out/r3/host_path_r3.log 2>&1 && cat gpurun_out/r3/host_path_r3.log
timeout -k 10 300 python bench.py --config 2 --steps 3 --warmup 1 > gpurun_out/r3/bench_config2.json 2> gpurun_out/r3/bench_config2.err && cat gpurun_out/r3/bench_config2.json
