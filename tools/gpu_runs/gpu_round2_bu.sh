#!/bin/bash
# round-end rehearsal: full GPU test suite, smoke(), default bench
set -o pipefail
mkdir -p gpurun_out/bu
export TMPDIR=/tmp DPAMD_HEARTBEAT_S=45
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/bu/tests.log 2>&1 || { tail -30 gpurun_out/bu/tests.log; exit 1; }
tail -1 gpurun_out/bu/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/bu/smoke.log 2>&1 || { tail -20 gpurun_out/bu/smoke.log; exit 1; }
tail -1 gpurun_out/bu/smoke.log
timeout -k 10 500 python bench.py > gpurun_out/bu/bench.log 2>&1; tail -1 gpurun_out/bu/bench.log | cut -c1-240
