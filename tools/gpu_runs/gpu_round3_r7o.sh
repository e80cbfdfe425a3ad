#!/bin/bash
# the full GPU suite six times in a row (ending at the first run that does not pass): is anything else in it occasional?
set -u
O=gpurun_out/r7o; mkdir -p $O
for i in 1 2 3 4 5 6; do
  s=$(date +%s)
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/run_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(( $(date +%s) - s )) s: $(tail -1 $O/run_$i.log | cut -c1-90)"
  [ $rc -ne 0 ] && { grep -v "Warning\|amdgpu.ids" $O/run_$i.log | tail -60 | cut -c1-220; exit 1; }
done
exit 0
