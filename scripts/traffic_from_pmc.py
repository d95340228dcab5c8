"""profiles/traffic.json from the two rocprofv3 --pmc passes of scripts/collect_profiles.sh.

FETCH_SIZE and WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE counts 128-byte
requests at 64 bytes, so it is doubled (the decoder pass checks that: its corrected fetch comes out at
the compressed input size).  Usage: python scripts/traffic_from_pmc.py <dir with pmc_fetch.csv, pmc_write.csv> <out.json>"""
import csv, json, sys
d, out = sys.argv[1], sys.argv[2]

def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for key in ("encode_fast_kernel", "decode_kernel"):
            if key in r["Kernel_Name"] and "redo" not in r["Kernel_Name"] and "order" not in r["Kernel_Name"]:
                acc.setdefault(key, []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}

f, w = per_kernel(f"{d}/pmc_fetch.csv", "FETCH_SIZE"), per_kernel(f"{d}/pmc_write.csv", "WRITE_SIZE")
res = {}
for k in f:
    fetch, write = int(f[k] * 1024 * 2), int(w[k] * 1024)
    res[k] = {"hbm_bytes_per_launch": fetch + write, "blocks_per_launch": 16384, "fetch_bytes_corrected": fetch, "write_bytes": write,
              "FETCH_SIZE_KiB_raw": f[k], "WRITE_SIZE_KiB_raw": w[k],
              "note": "rocprofv3 --pmc, separate passes over bench.py (16384 blocks/launch, scripts/collect_profiles.sh); "
                      "FETCH_SIZE (KiB) doubled per the gfx950 correction"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in res.items()}))
