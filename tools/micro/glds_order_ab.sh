cd $GRAFT_REPO_ROOT
bash tools/micro/conv_shape_pmc.sh 2>&1 | tail -16
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_next.py -q -k "conv" 2>&1 | tail -3
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['value_sync'], r['ms_per_step'], r['op_calls_per_step'])"; }
run chunk_outer
YMK_GLDS_TAP_OUTER=1 run tap_outer
run chunk_outer2
YMK_GLDS_TAP_OUTER=1 run tap_outer2
YMK_BENCH_CALLS=gpurun_out/t_calls4.log python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-sync-leg > /dev/null 2>&1; head -1 gpurun_out/t_calls4.log; grep "k3" gpurun_out/t_calls4.log | grep glds
