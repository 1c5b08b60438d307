#!/bin/bash
# GPU box (one gpurun call; the reference staged by tools/stage_reference.sh): the tests the driver's box skips (no reference checkout) on the
# FINAL tree of round 6 — the hooked reference on the GPU (predict detect / segment / half, validator, AutoBackend), the RCCL code path at
# world size 1 incl. the 200-step run — plus the bench line with the collective forced and the reference's own eager times.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
T0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_dropin_reference.py tests/test_gpu_bench_flow.py -m gpu -q -s > $O/r06_closures.log 2>&1
echo "closures rc=$? t=$(( $(date +%s) - T0 ))s" | tee -a $O/r06_closures.log
timeout 300 python bench.py --force-dist --steps 30 --warmup 10 --no-cpu-baseline > $O/r06_bench_force_dist.json 2> $O/r06_bench_force_dist.err
echo "force-dist rc=$? t=$(( $(date +%s) - T0 ))s"
if [ -d .refstage/ultralytics ]; then
  timeout 600 python tools/gpu_reference_timing.py s 64 r06 > $O/r06_reference_timing.log 2>&1
  echo "reference timing rc=$? t=$(( $(date +%s) - T0 ))s"
fi
tail -4 $O/r06_closures.log
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06_bench_force_dist.json") if l.startswith("{")][0])   # (RCCL prints its version banner after the line)
print("force-dist", r["value"], r.get("value_sync"), r["ms_per_step"], r.get("host_us_per_step_launch"), r["config"].get("collectives"))
PY
tail -2 $O/r06_reference_timing.log | head -c 3000
