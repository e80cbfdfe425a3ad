# the GPU suite under the remaining A/B knobs: folds off, auxiliary stream off, autotuning off (default tiles)
mkdir -p gpurun_out/r5v
run() { name=$1; shift; env "$@" timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 150 --timeout-method thread -p no:cacheprovider > gpurun_out/r5v/$name.log 2>&1; rc=$?; echo "$name rc=$rc: $(grep -c '^FAILED' gpurun_out/r5v/$name.log) failed; $(tail -1 gpurun_out/r5v/$name.log)"; grep '^FAILED' gpurun_out/r5v/$name.log | cut -c1-200 | head -8; [ $rc -eq 124 -o $rc -eq 137 ] && exit 1; return 0; }
run folds_off DPAMD_BN3_FOLD=0 DPAMD_SEPCONV_FOLD=0
run aux_off DPAMD_AUX_STREAM=0
run autotune_off DPAMD_CONV_AUTOTUNE=0
