#!/bin/bash
# SQ issue counters of one kernel of any command, averaged per launch.
# usage: bash scripts/pmc_sq_cmd.sh <tag> <kernel name part> <program> [args...]; output: gpurun_out/r3/pmc_issue_<tag>.log
export TMPDIR=/tmp
tag=$1; export PMC_KERNEL=$2; shift 2
out=gpurun_out/pmc_sq_cmd
mkdir -p gpurun_out/r3
pass() {
  c=$1; shift
  rm -rf $out && mkdir -p $out
  timeout -s KILL 240 rocprofv3 --pmc $c -d $out -o s -f csv -- "$@" > $out/log.txt 2>&1 || { tail -5 $out/log.txt; return 1; }
  python - <<'P'
import csv,collections,glob,os
f=glob.glob('gpurun_out/pmc_sq_cmd/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if os.environ['PMC_KERNEL'] in k:
        print(k, len(next(iter(v.values()))), 'launches', {c: round(sum(x)/len(x)) for c,x in v.items()})
P
}
{
pass "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "$@" &&
pass "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAVES" "$@"
} > gpurun_out/r3/pmc_issue_$tag.log 2>&1
cat gpurun_out/r3/pmc_issue_$tag.log
