#!/bin/bash
# L2 / L1 request counters over one corpus class (argument: class, blocks), two passes.
export TMPDIR=/tmp
for pass in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum" "TCC_CYCLE_sum TCC_WRITE_sum TCC_READ_sum TCC_ATOMIC_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum"; do
  out=gpurun_out/pmc_l2_cls
  rm -rf $out && mkdir -p $out
  timeout -s KILL 120 rocprofv3 --pmc $pass -d $out -o s -f csv -- python scripts/gpu_one_class.py $1 $2 > $out/log.txt 2>&1
  python - <<'P'
import csv,collections,glob
f=glob.glob('gpurun_out/pmc_l2_cls/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'lz4mi' in k and ('encode' in k or 'decode_kernel' in k):
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
P
done
