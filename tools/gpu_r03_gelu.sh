#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
T=${1:-r03u}
timeout -k 10 900 python -m pytest tests/test_gpu_next.py tests/test_gpu_mixture.py tests/test_gpu_model.py -m gpu -q --tb=short --no-header -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1
echo "tests: exit $?"; tail -3 gpurun_out/${T}_tests.log; grep -E "^(FAILED|ERROR)" gpurun_out/${T}_tests.log | head -20
for i in 1 2; do
python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('cfg5:', r['value'], r['ms_per_step'])"
python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('bench:', r['value'], r['ms_per_step'], [(f['kernel'][:30], f['ms_per_step']) for f in r['families'][:8] if 'glds' in f['kernel']])"
done
