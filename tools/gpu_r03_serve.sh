#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bench_flow.py -m gpu -q --tb=short --no-header -p no:cacheprovider > gpurun_out/r03g_flow.log 2>&1; echo "flow tests: exit $?"; tail -3 gpurun_out/r03g_flow.log
python tools/serve_bench.py > gpurun_out/r03g_serve.json 2> gpurun_out/r03g_serve.err; echo "serve: exit $?"; cat gpurun_out/r03g_serve.json; tail -3 gpurun_out/r03g_serve.err
python tools/serve_bench.py --serial > gpurun_out/r03g_serve_serial.json 2>> gpurun_out/r03g_serve.err; cat gpurun_out/r03g_serve_serial.json
python tools/serve_bench.py --frames 640x640 > gpurun_out/r03g_serve_640.json 2>> gpurun_out/r03g_serve.err; cat gpurun_out/r03g_serve_640.json
