R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R && python tools/micro/conv_shape_pmc.py run > /dev/null 2>&1 || { echo "workload crashes"; exit 1; }
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/cshape -- python $R/tools/micro/conv_shape_pmc.py run > $R/gpurun_out/cshape.log 2>&1
cd $R && python tools/micro/conv_shape_pmc.py report $(ls gpurun_out/cshape/*/*.db | head -1) | tee gpurun_out/conv_shape_fetch.txt
rm -rf gpurun_out/cshape
