mkdir -p gpurun_out/r3z && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r3z/$name.log 2>&1; echo "$name rc=$?"; }
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_hardening.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3z/pytest.log 2>&1; tail -3 gpurun_out/r3z/pytest.log
run r101_b256 python bench.py --batch 256 --steps 12 --warmup 5
run r101_b256_2 python bench.py --batch 256 --steps 12 --warmup 5
run r101_b384 python bench.py --batch 384 --steps 10 --warmup 4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3z/prof_b256 -o p -- python3 bench.py --steps 4 --warmup 3 --batch 256 > gpurun_out/r3z/prof_b256.log 2>&1
rm -f gpurun_out/r3z/*/*kernel_trace.csv
grep -o "\"value\": [0-9.]*\|hbm_reserved[a-z_]*\": [0-9.]*" gpurun_out/r3z/*.log
grep "upsample_ce" gpurun_out/r3z/prof_b256/p_kernel_stats.csv | cut -c1-200
