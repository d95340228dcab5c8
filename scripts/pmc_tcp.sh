#!/bin/bash
# Vector-memory path of the encode and decode launch on one corpus class: address translation,
# L1 -> L2 request latency, texture-addresser occupancy.  Arguments: class, blocks.
# Each counter pass is a run of its own.  Output: gpurun_out/pmc_tcp/
export TMPDIR=/tmp
out=gpurun_out/pmc_tcp
rm -rf $out && mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE -d $out/a -o s -f csv -- python scripts/gpu_one_class.py $1 $2 > $out/a.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_FLAT_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -d $out/b -o s -f csv -- python scripts/gpu_one_class.py $1 $2 > $out/b.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum -d $out/c -o s -f csv -- python scripts/gpu_one_class.py $1 $2 > $out/c.log 2>&1
python - <<'P'
import csv,collections,glob
for f in sorted(glob.glob('gpurun_out/pmc_tcp/**/*counter_collection.csv',recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'lz4mi' in k:
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
P
