#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/bw
cd /tmp && export TMPDIR=/tmp
timeout -k 5 60 rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/bw/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $GRAFT_REPO_ROOT/gpurun_out/bw/counters.txt | sort -u | grep -i "VMEM\|WAIT\|BARRIER\|TEX\|LDS_DATA\|INST_CYCLES\|ISSUE" | tr '\n' ' '
