#!/bin/bash
# r04 call K: the full-size graph parity test failed once (r04i, max-normalised error 0.079): repeat it in fresh processes, keep each
# process's tile choices
OUT=$PWD/gpurun_out/r04k
mkdir -p $OUT
for n in 1 2 3 4; do
  AA_DUMP_TILE_CACHE=$OUT/tiles_$n.json timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "test_unet_forward_at_the_metric_configuration and not bf16" > $OUT/run_$n.log 2>&1; echo "run $n rc=$?" >> $OUT/summary.log
  grep -a "AssertionError: \|passed\|failed" $OUT/run_$n.log | head -3
done
cat $OUT/summary.log
