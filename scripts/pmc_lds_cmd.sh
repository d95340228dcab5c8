#!/bin/bash
# LDS counters of one kernel of any command, averaged per launch.
# usage: bash scripts/pmc_lds_cmd.sh <tag> <kernel name part> <program> [args...]; output: gpurun_out/r3/pmc_lds_<tag>.log
export TMPDIR=/tmp
tag=$1; export PMC_KERNEL=$2; shift 2
out=gpurun_out/pmc_lds_cmd
mkdir -p gpurun_out/r3
rm -rf $out && mkdir -p $out
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -d $out -o s -f csv -- "$@" > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python - <<'P' > gpurun_out/r3/pmc_lds_$tag.log
import csv,collections,glob,os
f=glob.glob('gpurun_out/pmc_lds_cmd/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if os.environ['PMC_KERNEL'] in k:
        print(k, len(next(iter(v.values()))), 'launches', {c: round(sum(x)/len(x)) for c,x in v.items()})
P
cat gpurun_out/r3/pmc_lds_$tag.log
