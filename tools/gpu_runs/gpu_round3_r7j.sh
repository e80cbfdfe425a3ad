#!/bin/bash
# kernel breakdown of the final tree's training step (bench.py default: R101 513^2 B256) and of the Xception-65 / DANet defaults
set -u
O=$PWD/gpurun_out/r7j; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
prof() { name=$1; steps=$2; title=$3; shift 3
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o p -- python3 $R/bench.py --steps 4 --warmup 3 "$@" > $O/$name.log 2>&1 || { tail -5 $O/$name.log; exit 1; }
  grep -o '"value": [0-9.]*' $O/$name.log
  python $R/tools/summarize_prof.py $O/prof_$name/p_kernel_stats.csv $steps "$title" > $O/${name}_profile.md 2>&1
  head -16 $O/${name}_profile.md
}
prof r101_b256 7 "R101 513^2 B256 v55 (3 warmup + 4 timed steps)"
prof x65_b128 7 "Xception-65 OS8 B128 v55 (3 warmup + 4 timed steps)" --model deeplabv3plus_xception65 --output-stride 8
prof danet_b64 7 "DANet-R101 512^2 instance task B64 v55 (3 warmup + 4 timed steps)" --task instance --model danet_resnet101 --crop 512
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete
