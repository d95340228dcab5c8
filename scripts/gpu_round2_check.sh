#!/bin/bash
# GPU tests, then the bench lines (headline, one-rank rehearsal of the sharded step, configs 2 and 4).
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py > gpurun_out/bench_r2.json 2> gpurun_out/bench_r2.err || { tail -20 gpurun_out/bench_r2.err; exit 1; }
LZ4MI_BENCH_DIST=1 timeout -k 10 300 python bench.py > gpurun_out/bench_r2_dist1.json 2> gpurun_out/bench_r2_dist1.err || { tail -20 gpurun_out/bench_r2_dist1.err; exit 1; }
timeout -k 10 400 python bench.py --config 2 --steps 3 --warmup 1 > gpurun_out/bench_r2_config2.json 2> gpurun_out/bench_r2_config2.err || { tail -20 gpurun_out/bench_r2_config2.err; exit 1; }
timeout -k 10 400 python bench.py --config 4 --steps 3 --warmup 1 > gpurun_out/bench_r2_config4.json 2> gpurun_out/bench_r2_config4.err || { tail -20 gpurun_out/bench_r2_config4.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_r2", "bench_r2_dist1", "bench_r2_config2", "bench_r2_config4"):
    d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), d["unit"], {k: round(v, 3) for k, v in d.get("phase_ms", {}).items()},
          {k: round(v["frac"], 4) for k, v in d["roofline_all"].items()})
PY
