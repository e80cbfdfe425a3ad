#!/bin/bash
# where the Xception-65 (output stride 16, bench_eval default) and DANet evaluation forwards spend their time
set -u
O=$PWD/gpurun_out/r7a; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for m in deeplabv3plus_xception65 danet_resnet101; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o p -- python3 $R/tools/bench_eval.py $m 16 > $O/$m.log 2>&1 || { tail -5 $O/$m.log; exit 1; }
grep "inference:" $O/$m.log
python $R/tools/summarize_prof.py $O/prof_$m/p_kernel_stats.csv 72 "$m 513^2 evaluation forward, batch 16 (35 eager + 2 capture warm-ups + 35 replays; tile measurements of the first forward included)" > $O/${m}_profile.md 2>&1
head -32 $O/${m}_profile.md
done
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete
