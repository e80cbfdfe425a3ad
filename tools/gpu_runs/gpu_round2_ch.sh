#!/bin/bash
# weight-gradient kernel: s_setprio 1 around the MFMA groups (diagnostic build -DDP_WG_PRIO=1) against the default, block stamps
set -o pipefail
mkdir -p gpurun_out/ch
S="256:1024:33:1 1024:256:33:1 256:256:33:3 512:2048:33:1 2048:256:33:3:12"
timeout -k 10 300 python -u tools/stamp_conv.py --wgrad --batch 128 $S > gpurun_out/ch/default.txt 2>&1 &&
timeout -k 10 300 python -u tools/stamp_conv.py --wgrad --batch 128 --lib distributedpytorch_amd/csrc/build/libdpamd_stamp_prio.so $S > gpurun_out/ch/prio.txt 2>&1 &&
timeout -k 10 300 python -u tools/stamp_conv.py --wgrad --batch 128 $S > gpurun_out/ch/default_b.txt 2>&1
for f in default prio default_b; do echo "## $f"; grep -v amdgpu gpurun_out/ch/$f.txt; done
