cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['value_sync'], r['ms_per_step'], r['p50_batch_ms_sync'])"; }
for i in 1 2; do run default; YMK_ENABLE=32 run early_levels; YMK_ENABLE=16 run level_streams; done
