#!/bin/bash
# same-box A/B of the whole tree: HEAD against the tree of v51 (b0e2668, before predict / evaluation epilogue / p2p change)
set -u
O=$PWD/gpurun_out/r7i; mkdir -p $O
for i in 1 2 3; do
  (cd _ab_v51 && timeout -k 10 300 python bench.py > $O/r101_v51_$i.log 2>&1) || { tail -5 $O/r101_v51_$i.log; exit 1; }
  timeout -k 10 300 python bench.py > $O/r101_head_$i.log 2>&1 || { tail -5 $O/r101_head_$i.log; exit 1; }
done
grep -o '"value": [0-9.]*' $O/r101_*.log | sed "s|$O/||"
