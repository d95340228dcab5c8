#!/bin/bash
# Memory traffic and the vector-memory path of the HC encode launch on one corpus class.  Arguments: class, blocks.
# Each counter pass is a run of its own.  Output: gpurun_out/pmc_hc/
export TMPDIR=/tmp
out=gpurun_out/pmc_hc
rm -rf $out && mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/a -o s -f csv -- python scripts/gpu_one_class_hc.py $1 $2 > $out/a.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/b -o s -f csv -- python scripts/gpu_one_class_hc.py $1 $2 > $out/b.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE -d $out/c -o s -f csv -- python scripts/gpu_one_class_hc.py $1 $2 > $out/c.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $out/d -o s -f csv -- python scripts/gpu_one_class_hc.py $1 $2 > $out/d.log 2>&1
python - <<'P'
import csv,collections,glob
for f in sorted(glob.glob('gpurun_out/pmc_hc/**/*counter_collection.csv',recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'lz4mi' in k:
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
P
