#!/bin/bash
# The configs[3] per-GPU shard (131072 blocks = 8 GiB) on one GPU: block round trip, then the sharded step with one rank.
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --blocks 131072 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r2_8gib.json 2> gpurun_out/bench_r2_8gib.err || { tail -5 gpurun_out/bench_r2_8gib.err; exit 1; }
LZ4MI_BENCH_DIST=1 timeout -k 10 500 python bench.py --corpus-gib 8 --steps 5 --warmup 1 > gpurun_out/bench_r2_8gib_dist1.json 2> gpurun_out/bench_r2_8gib_dist1.err || { tail -5 gpurun_out/bench_r2_8gib_dist1.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_r2_8gib", "bench_r2_8gib_dist1"):
    d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), d["unit"], {k: round(v, 3) for k, v in d.get("phase_ms", {}).items()},
          {k: round(v["frac"], 4) for k, v in d["roofline_all"].items()})
PY
