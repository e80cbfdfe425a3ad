# flake check of the multi-process tests (the driver runs the suite once with -x): four passes, stop at the first failure
mkdir -p gpurun_out/r5i
for i in 1 2 3 4; do
  timeout -k 10 400 python -u -m pytest tests/test_gpu_p2p.py tests/test_gpu_runtime.py -x -q --timeout 200 --timeout-method thread -p no:cacheprovider > gpurun_out/r5i/pass_$i.log 2>&1; rc=$?
  echo "pass $i rc=$rc: $(tail -1 gpurun_out/r5i/pass_$i.log)"
  [ $rc -eq 0 ] || { tail -30 gpurun_out/r5i/pass_$i.log | cut -c1-200; exit 1; }
done
