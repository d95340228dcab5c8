#!/bin/bash
# Cross-check of the FETCH_SIZE / WRITE_SIZE figures for launches whose traffic is dword gathers and scattered
# stores: the L2's memory-side request counters (32-byte and 64-byte requests) next to the derived byte counters.
# Arguments: class, blocks.  Each counter pass is a run of its own.
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum"; do
  out=gpurun_out/pmc_ea
  rm -rf $out && mkdir -p $out
  timeout -s KILL 120 rocprofv3 --pmc $pass -d $out -o s -f csv -- python scripts/gpu_one_class.py $1 $2 > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
  python - <<'P'
import csv,collections,glob
f=glob.glob('gpurun_out/pmc_ea/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'lz4mi' in k and ('encode' in k or 'decode_kernel' in k):
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
P
done
