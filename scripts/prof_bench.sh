#!/bin/bash
# Kernel-time summary of a short bench run.  Output: gpurun_out/prof_bench/kernel_stats.csv
export TMPDIR=/tmp
out=gpurun_out/prof_bench
rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -f csv -- python bench.py --no-cpu-baseline --steps 8 --warmup 2 > $out/trace.log 2>&1 || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
cut -d, -f1-7 $out/kernel_stats.csv | head -12
