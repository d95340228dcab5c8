#!/bin/bash
# Second part of the round's profile set (a call of its own: the two parts together pass one call's time limit):
# the 64 GiB single-GPU run of configs[3]'s per-GPU share, and the host-buffer frame calls.
set -e
out=gpurun_out/profiles_new
mkdir -p $out
# host-buffer frame calls through the C ABI (256 MiB): independent blocks with the content checksum, linked blocks
# all at once and on the serial chain
timeout -k 10 200 python scripts/gpu_host_path_timing.py > $out/host_path.log 2>&1
timeout -k 10 200 python scripts/gpu_host_path_timing.py ck >> $out/host_path.log 2>&1
timeout -k 10 200 python scripts/gpu_host_path_timing.py linked >> $out/host_path.log 2>&1
LZ4MI_LINK_RESOLVE=0 timeout -k 10 200 python scripts/gpu_host_path_timing.py linked >> $out/host_path.log 2>&1
# (the 64 GiB run goes last: the host holds 64 GiB of corpus for it, and pageable copies measured right behind it
# run at half their rate)
timeout -k 10 900 python bench.py --gpus 1 --corpus-gib 64 --steps 3 --warmup 1 > $out/bench_64gib.json 2> $out/bench_64gib.err
ls -la $out
