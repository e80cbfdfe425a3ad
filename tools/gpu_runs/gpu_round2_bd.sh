#!/bin/bash
# which small ATen ops / device copies a native Xception-65 step still launches
set -o pipefail
mkdir -p gpurun_out/bd
timeout -k 10 300 python -u tools/count_small_ops.py deeplabv3plus_xception65 > gpurun_out/bd/small_ops_x65.log 2>&1; grep -v amdgpu gpurun_out/bd/small_ops_x65.log | tail -90 | cut -c1-250
