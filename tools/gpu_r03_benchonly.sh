#!/bin/bash
# bench lines only (the counter summaries of the current sources are already under profiles/)
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python bench.py --steps 30 --warmup 10 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench: exit $?"
python bench.py --steps 30 --warmup 10 --dtype f16 --no-cpu-baseline > gpurun_out/${TAG}_bench_f16.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --dtype f32 --no-cpu-baseline > gpurun_out/${TAG}_bench_f32.json 2>/dev/null
python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg5.json 2>/dev/null
python tools/serve_bench.py > gpurun_out/${TAG}_serve.json 2> gpurun_out/${TAG}_serve.err
python - <<PY
import json
for n in ("bench", "bench_f16", "bench_f32", "bench_cfg5", "serve"):
    try:
        r = json.loads(open("gpurun_out/${TAG}_%s.json" % n).read())
        print(n, r["value"], r["ms_per_step"], (r.get("roofline") or {}).get("kernel"), (r.get("roofline") or {}).get("frac"), (r.get("roofline") or {}).get("traffic"))
    except Exception as e:
        print(n, "unreadable", e)
PY
