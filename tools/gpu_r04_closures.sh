#!/bin/bash
# GPU box (one gpurun call; the reference staged by tools/stage_reference.sh): the configurations VERDICT round 3 listed as never run —
# the hooked reference on the GPU, the RCCL code path at world size 1, config 5 as stated, segment NMS — plus the GPU suite and the
# bench line with value_sync.  Everything lands in gpurun_out/r04a_*.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_dropin_reference.py tests/test_gpu_bench_flow.py -m gpu -q -s > $O/r04a_closures.log 2>&1
echo "closures rc=$? t=$(( $(date +%s) - T0 ))s" | tee -a $O/r04a_closures.log
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_dropin_reference.py --deselect tests/test_gpu_bench_flow.py > $O/r04a_gpu_suite.log 2>&1
echo "suite rc=$? t=$(( $(date +%s) - T0 ))s" | tee -a $O/r04a_gpu_suite.log
timeout 300 python bench.py --steps 30 --warmup 10 > $O/r04a_bench.json 2> $O/r04a_bench.err
echo "bench rc=$? t=$(( $(date +%s) - T0 ))s"
timeout 300 python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --dtype f16 --cluster --sigma 0.1 --dense --imbalance 8,3 --steps 10 --warmup 3 > $O/r04a_bench_cfg5.json 2> $O/r04a_bench_cfg5.err
echo "cfg5 rc=$? t=$(( $(date +%s) - T0 ))s"
timeout 300 python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --dtype f16 --cluster --sigma 0.1 --steps 10 --warmup 3 > $O/r04a_bench_cfg5_balanced.json 2> $O/r04a_bench_cfg5_balanced.err
echo "cfg5 balanced rc=$? t=$(( $(date +%s) - T0 ))s"
if [ -d .refstage/ultralytics ]; then
  timeout 600 python tools/gpu_reference_timing.py s 64 r04 > $O/r04a_reference_timing.log 2>&1
  echo "reference timing rc=$? t=$(( $(date +%s) - T0 ))s"
fi
tail -3 $O/r04a_closures.log $O/r04a_gpu_suite.log
head -c 1500 $O/r04a_bench.json; echo
head -c 600 $O/r04a_bench_cfg5.json; echo
tail -2 $O/r04a_reference_timing.log | head -c 3000
