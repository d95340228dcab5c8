#!/bin/bash
# Instruction mix and issue-unit activity of the encode and decode launch on one corpus class.
# Arguments: class, blocks.  Two counter passes, each a run of its own.  Output: gpurun_out/pmc_mix/
export TMPDIR=/tmp
out=gpurun_out/pmc_mix
rm -rf $out && mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES -d $out/a -o s -f csv -- python scripts/gpu_one_class.py $1 $2 > $out/a.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $out/b -o s -f csv -- python scripts/gpu_one_class.py $1 $2 > $out/b.log 2>&1
python - <<'P'
import csv,collections,glob
for f in sorted(glob.glob('gpurun_out/pmc_mix/**/*counter_collection.csv',recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'lz4mi' in k:
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
P
