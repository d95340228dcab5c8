#!/bin/bash
# SQ counter pass over one corpus class (argument: class, blocks).  Output: gpurun_out/pmc_sq_cls/
export TMPDIR=/tmp
out=gpurun_out/pmc_sq_cls
rm -rf $out && mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY} -d $out -o s -f csv -- python scripts/gpu_one_class.py $1 $2 > $out/log.txt 2>&1
python - <<'P'
import csv,collections,glob
f=glob.glob('gpurun_out/pmc_sq_cls/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'lz4mi' in k:
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
P
