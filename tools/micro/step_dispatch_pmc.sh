R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --split 1 --pipeline 1 --no-sync-leg"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/sd_$C -- $BENCH > $R/gpurun_out/sd_$C.log 2>&1
done
cd $R
python tools/micro/step_dispatch_pmc.py $(ls gpurun_out/sd_FETCH_SIZE/*/*.db | head -1) $(ls gpurun_out/sd_WRITE_SIZE/*/*.db | head -1) > gpurun_out/step_dispatch_pmc.txt 2>&1
rm -rf gpurun_out/sd_FETCH_SIZE gpurun_out/sd_WRITE_SIZE
YMK_BENCH_CALLS=gpurun_out/step_calls.log python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-sync-leg > /dev/null 2>&1
tail -3 gpurun_out/step_dispatch_pmc.txt
