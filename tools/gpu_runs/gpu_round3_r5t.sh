# split-K counts of the heaviest weight gradients at B256 (0 = what the step uses: table / autotune)
mkdir -p gpurun_out/r5t
timeout -k 10 300 python -u tools/bench_conv.py --batch 256 --filter aspp.d --splits 0,1,2,3,4,5,6,8,12 > gpurun_out/r5t/aspp.log 2>&1 && \
timeout -k 10 300 python -u tools/bench_conv.py --batch 256 --filter dec.last --splits 0,1,2,3,4,5,6,8,12 > gpurun_out/r5t/dec.log 2>&1 && \
timeout -k 10 300 python -u tools/bench_conv.py --batch 256 --filter l3.conv2 --splits 0,1,2,3,4,5,6,8,12 > gpurun_out/r5t/l3.log 2>&1; rc=$?
grep -h "^layer\|aspp\|dec\|l3" gpurun_out/r5t/*.log | grep -v amdgpu | cut -c1-160; exit $rc
