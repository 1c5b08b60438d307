#!/bin/bash
# Round-5 first GPU call: LDS-form A/B of the depthwise stencil, per-call A/B of library variants, bench line on both weight sets, kernel tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python tools/micro/dw_lds_ab.py 5 > gpurun_out/r05_dw_lds_ab.txt 2>&1; echo "dw_lds_ab: exit $?"; cat gpurun_out/r05_dw_lds_ab.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "dw or depthwise or esmoe or detect or stencil" > gpurun_out/r05a_kernel_tests.log 2>&1; echo "kernel tests: exit $?"; tail -3 gpurun_out/r05a_kernel_tests.log
python bench.py --steps 40 --warmup 10 > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; echo "bench: exit $?"; head -c 600 gpurun_out/r05a_bench.json; echo; tail -3 gpurun_out/r05a_bench.err
python bench.py --steps 40 --warmup 10 --weights recipe --no-cpu-baseline > gpurun_out/r05a_bench_recipe.json 2>/dev/null; echo "bench recipe: exit $?"; head -c 300 gpurun_out/r05a_bench_recipe.json; echo
bash tools/micro/calls_ab.sh "moe_dw|dwconv|detect_cls|conv_glds_kernel<(256|128)" 2 dwmode0=tools/micro/_dwab/libymk_dwmode0.so dcserial=tools/micro/_dwab/libymk_dcserial.so setprio=tools/micro/_dwab/libymk_setprio.so > gpurun_out/r05a_calls_ab.txt 2>&1; echo "calls_ab: exit $?"
grep -E "^round|^---" gpurun_out/r05a_calls_ab.txt
python - <<'PY'
import json
for n in ("r05a_bench", "r05a_bench_recipe"):
    try:
        r = json.loads(open(f"gpurun_out/{n}.json").read())
        print(n, r["value"], r["value_sync"], r["ms_per_step"], r["retained_pairs"], r.get("roofline_step"), {k: (v["ms_per_step_eager"], v["hbm_frac"], v["mfma_frac"]) for k, v in (r.get("roofline_layers") or {}).items()})
    except Exception as e:
        print(n, "unreadable", e)
PY
