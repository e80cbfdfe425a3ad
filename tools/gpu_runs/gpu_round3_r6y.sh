#!/bin/bash
# where the evaluation forward's time goes (R101 513^2 batch 32 and batch 1)
set -u
O=$PWD/gpurun_out/r6y; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b32 -o b32 -- python3 $R/tools/bench_eval.py deeplabv3plus_resnet101 32 > $O/b32.log 2>&1 || { tail -5 $O/b32.log; exit 1; }
grep "inference:" $O/b32.log
cd $R
f=$(ls $O/prof_b32/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls $O/prof_b32/*kernel_stats.csv | head -1)
python tools/summarize_prof.py $f 72 "R101 513^2 evaluation forward, batch 32 (35 eager + 2 capture warm-ups + 35 replays)" > $O/b32_profile.md 2>&1 || head -30 $f > $O/b32_profile.md
head -40 $O/b32_profile.md
timeout -k 10 200 python -u tools/bench_eval.py deeplabv3plus_resnet101 8 > $O/b8.log 2>&1; tail -1 $O/b8.log
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
