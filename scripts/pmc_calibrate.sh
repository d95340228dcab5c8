#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per access for random dword gathers, random dword stores and a streaming read
# (scripts/micro/gather_fetch.hip, built beforehand).  Each counter pass is a run of its own.
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  out=gpurun_out/pmc_cal
  rm -rf $out && mkdir -p $out
  timeout -s KILL 120 rocprofv3 --pmc $pass -d $out -o s -f csv -- scripts/micro/gather_fetch > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
  grep -E "^(gather|scatter|stream):" $out/log.txt | tail -3
  python3 - <<'P'
import csv,collections,glob
f=glob.glob('gpurun_out/pmc_cal/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:24]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
P
done
