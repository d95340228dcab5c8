#!/bin/bash
# SQ issue counters over the 1 GiB corpus launch of the fast encoder (scripts/gpu_enc_only.py).
# usage: bash scripts/pmc_sq_corpus.sh <tag>; output: gpurun_out/r3/pmc_issue_<tag>.log
export TMPDIR=/tmp
tag=${1:-cur}
out=gpurun_out/pmc_sq_corpus
mkdir -p gpurun_out/r3
pass() {
  rm -rf $out && mkdir -p $out
  timeout -s KILL 180 rocprofv3 --pmc $1 -d $out -o s -f csv -- python scripts/gpu_enc_only.py > $out/log.txt 2>&1 || { tail -5 $out/log.txt; return 1; }
  python - <<'P'
import csv,collections,glob
f=glob.glob('gpurun_out/pmc_sq_corpus/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'encode_fast' in k:
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
P
}
{
pass "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" &&
pass "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAVES"
} > gpurun_out/r3/pmc_issue_$tag.log 2>&1
cat gpurun_out/r3/pmc_issue_$tag.log
