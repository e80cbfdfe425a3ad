# the stall reproducer at the new default (normal priority), six fresh processes in a row; stops at the first stall
mkdir -p gpurun_out/r4v
for i in 1 2 3 4 5 6; do
  timeout -k 10 120 python -u tools/diag_hang.py > gpurun_out/r4v/run_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(grep -c 'loader test passed' gpurun_out/r4v/run_$i.log)"
  [ $rc -eq 0 ] || exit 1
done
