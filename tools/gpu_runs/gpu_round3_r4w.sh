# the GPU suite with the test files in reverse order (state leaking between tests shows up as order dependence)
mkdir -p gpurun_out/r4w
FILES=$(ls tests/test_*.py | sort -r | tr '\n' ' ')
timeout -k 10 900 python -u -m pytest $FILES -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/r4w/pytest_reverse.log 2>&1; rc=$?
tail -5 gpurun_out/r4w/pytest_reverse.log | cut -c1-300; exit $rc
