cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "detect or decode" 2>&1 | tail -3
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['value_sync'], r['ms_per_step'], r['op_calls_per_step'])"; }
run fused
YMK_DISABLE=4194304 run unfused
run fused2
YMK_DISABLE=4194304 run unfused2
YMK_BENCH_CALLS=gpurun_out/t_calls3.log python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-sync-leg > /dev/null 2>&1; head -1 gpurun_out/t_calls3.log; grep -i "detect\|nms" gpurun_out/t_calls3.log
