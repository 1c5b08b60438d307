#!/bin/bash
# Round-6 evidence run (GPU box): kernel-trace stats, HBM-traffic PMC passes and SQ counter passes of the bench workload on the CURRENT
# kernel sources (the summaries carry their source hash), the bench line (value + value_sync), config 5 AS STATED (fp16 + CW-NMS + dense-scene
# NMS settings + expert imbalance) and balanced, the other two compute types, per-call logs, the kernel trace of the timed (pipelined) configuration.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
bash tools/gpu_profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1; echo "profile: exit $?"; tail -3 gpurun_out/${TAG}_profile.log
bash tools/gpu_pmc.sh $TAG "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE" > gpurun_out/${TAG}_sq.log 2>&1; echo "sq: exit $?"
cd /tmp && export TMPDIR=/tmp
CFG5="--cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --dtype f16 --cluster --sigma 0.1 --dense --imbalance 8,3"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_cfg5_trace -- python $R/bench.py $CFG5 --steps 3 --warmup 1 --no-cpu-baseline --no-graph --split 1 --pipeline 1 --no-sync-leg > $R/gpurun_out/${TAG}_cfg5_trace.log 2>&1
python $R/tools/prof_summary.py $(ls $R/gpurun_out/${TAG}_cfg5_trace/*/*.db | head -1) 9 > $R/gpurun_out/${TAG}_cfg5_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/${TAG}_cfg5_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_gtrace -- python $R/bench.py --steps 200 --warmup 6 --no-cpu-baseline --no-sync-leg > $R/gpurun_out/${TAG}_gtrace.log 2>&1
( echo "# rocprofv3 --kernel-trace of bench.py --steps 200 --warmup 6 --no-sync-leg (hipGraph replays, three batches in flight);"
  echo "# per-step columns divide by the replayed steps + warm-up / spin-up / diagnostic steps (about 280)"
  python $R/tools/prof_summary.py $(ls $R/gpurun_out/${TAG}_gtrace/*/*.db | head -1) 280 ) > $R/gpurun_out/${TAG}_graph_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/${TAG}_gtrace
cd $R
# the bench line quotes counter numbers only from summaries under profiles/ whose source hash is this tree's: place the fresh ones there
cp gpurun_out/${TAG}_pmc_FETCH_SIZE.json gpurun_out/${TAG}_pmc_WRITE_SIZE.json gpurun_out/${TAG}_sq_1.json gpurun_out/${TAG}_sq_2.json profiles/ 2>/dev/null
python bench.py $CFG5 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg5.json 2>/dev/null; echo "cfg5 as stated: exit $?"
python bench.py --cfg yolo-master-moa-mot.yaml --scale l --imgsz 1280 --batch 16 --dtype f16 --cluster --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg5_balanced.json 2>/dev/null
bash tools/micro/step_dispatch_pmc.sh > gpurun_out/${TAG}_step_dispatch.log 2>&1; cp gpurun_out/step_dispatch_pmc.txt gpurun_out/${TAG}_step_dispatch_pmc.txt; cp gpurun_out/${TAG}_step_dispatch_pmc.txt profiles/ 2>/dev/null
python tools/sq_summary.py ${TAG} > gpurun_out/${TAG}_sq_summary.txt 2>&1
python bench.py --steps 30 --warmup 10 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench: exit $?"; head -c 400 gpurun_out/${TAG}_bench.json; echo
python bench.py --steps 30 --warmup 10 --weights recipe --no-cpu-baseline > gpurun_out/${TAG}_bench_recipe_weights.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_args.json 2>/dev/null
python bench.py --steps 30 --warmup 10 --dtype f16 --no-cpu-baseline > gpurun_out/${TAG}_bench_f16.json 2>/dev/null; echo "bench f16: exit $?"
python bench.py --steps 10 --warmup 3 --dtype f32 --no-cpu-baseline > gpurun_out/${TAG}_bench_f32.json 2>/dev/null; echo "bench f32: exit $?"
YMK_BENCH_CALLS=gpurun_out/${TAG}_calls.log python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-sync-leg > /dev/null 2>&1
YMK_BENCH_CALLS=gpurun_out/${TAG}_cfg5_calls.log python bench.py $CFG5 --steps 3 --warmup 1 --no-cpu-baseline --no-sync-leg > /dev/null 2>&1
python tools/serve_bench.py > gpurun_out/${TAG}_serve.json 2> gpurun_out/${TAG}_serve.err; echo "serve: exit $?"
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gpu_suite.log 2>&1; echo "gpu suite: exit $?"; tail -n 2 gpurun_out/${TAG}_gpu_suite.log
python - <<PY
import json
for n in ("bench", "bench_recipe_weights", "bench_driver_args", "bench_f16", "bench_f32", "bench_cfg5", "bench_cfg5_balanced"):
    try:
        r = json.loads(open("gpurun_out/${TAG}_%s.json" % n).read())
        print(n, r["value"], r.get("value_sync"), r["ms_per_step"], r["roofline"]["kernel"], r["roofline"]["frac"], r["roofline"].get("traffic"))
    except Exception as e:
        print(n, "unreadable", e)
PY
