cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['value_sync'], r['ms_per_step'])"; }
run base
YMK_GLDS_SMALL_BELOW=100000000 run small128
YMK_GLDS_BIG_MIN_TILES=100000000 run nobig
YMK_GLDS_SMALL_BELOW=100000000 YMK_GLDS_BIG_MIN_TILES=100000000 run both
run base2
