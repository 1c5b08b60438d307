#!/bin/bash
# GPU box: the fused ES-MoE kernel — parity tests, then bench A/B (fused / two-kernel form) with per-call logs.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "esmoe" > $O/r04b_esf_tests.log 2>&1; echo "tests rc=$?"; tail -n 5 $O/r04b_esf_tests.log
YMK_BENCH_CALLS=$O/r04b_calls_fused.log timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/r04b_bench_fused.json 2> $O/r04b_bench_fused.err; echo "bench fused rc=$?"
YMK_DISABLE=2097152 YMK_BENCH_CALLS=$O/r04b_calls_unfused.log timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/r04b_bench_unfused.json 2> $O/r04b_bench_unfused.err; echo "bench unfused rc=$?"
python - <<'PY'
import json
for t in ("fused","unfused"):
    try:
        d=json.load(open(f"gpurun_out/r04b_bench_{t}.json"))
        print(t, d["value"], d["value_sync"], d["p50_batch_ms_sync"], [(f["kernel"][:12], f["ms_per_step"]) for f in d["families"] if "moe" in f["kernel"]])
    except Exception as e: print(t, "failed", e)
PY
grep "moe" $O/r04b_calls_fused.log | head -12
