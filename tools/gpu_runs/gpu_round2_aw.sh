#!/bin/bash
# which small ATen ops / device copies a native step still launches
set -o pipefail
mkdir -p gpurun_out/aw
timeout -k 10 300 python -u tools/count_small_ops.py > gpurun_out/aw/small_ops.log 2>&1; grep -v amdgpu gpurun_out/aw/small_ops.log | tail -75
