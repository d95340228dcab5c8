#!/bin/bash
# Round profile set: bench JSON, rocprofv3 kernel stats, and PMC traffic in two separate passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Run on the GPU box from the repo root;
# results land in gpurun_out/profiles_new/ for copying into profiles/.
set -e
out=gpurun_out/profiles_new
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -f csv -- python bench.py --no-cpu-baseline --no-by-class --steps 12 --warmup 1 > $out/trace.log 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o f -f csv -- python bench.py --no-cpu-baseline --no-by-class --steps 3 --warmup 1 > $out/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o w -f csv -- python bench.py --no-cpu-baseline --no-by-class --steps 3 --warmup 1 > $out/pmc_write.log 2>&1
cp $(find $out/pmc_fetch -name "*counter_collection.csv" | head -1) $out/pmc_fetch.csv
cp $(find $out/pmc_write -name "*counter_collection.csv" | head -1) $out/pmc_write.csv
rm -rf $out/trace $out/pmc_fetch $out/pmc_write
python scripts/traffic_from_pmc.py $out $out/traffic.json
# the other bench lines of the round: one-rank rehearsal of the sharded step, configs 2 and 4, the 8 GiB shard
LZ4MI_BENCH_DIST=1 timeout -k 10 300 python bench.py > $out/bench_dist1.json 2> $out/bench_dist1.err
timeout -k 10 400 python bench.py --config 2 --steps 8 --warmup 1 > $out/bench_config2.json 2> $out/bench_config2.err
timeout -k 10 400 python bench.py --config 4 --steps 3 --warmup 1 > $out/bench_config4.json 2> $out/bench_config4.err
timeout -k 10 500 python bench.py --blocks 131072 --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_8gib.json 2> $out/bench_8gib.err
LZ4MI_BENCH_DIST=1 timeout -k 10 500 python bench.py --corpus-gib 8 --steps 5 --warmup 1 > $out/bench_8gib_dist1.json 2> $out/bench_8gib_dist1.err
ls -la $out
