mkdir -p gpurun_out/r4g && cd $GRAFT_REPO_ROOT && \
timeout -k 10 200 python -u tools/bench_conv.py --batch 256 --cfgs 1,3,4,5,8,9,10,14,16 --filter l3.conv > gpurun_out/r4g/l3_cfgs.log 2>&1; rc=$?; grep -v amdgpu gpurun_out/r4g/l3_cfgs.log; exit $rc
