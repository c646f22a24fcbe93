#!/bin/bash
OUT=gpurun_out/r03q
mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -s -k "groupnorm" > $OUT/test_gn.log 2>&1; echo "gn tests rc=$?" >> $OUT/summary.log
tail -12 $OUT/test_gn.log
timeout 600 python scripts/bench_kernels.py --only "groupnorm" > $OUT/bench_norm.log 2>&1; echo "norm bench rc=$?" >> $OUT/summary.log
grep -v amdgpu $OUT/bench_norm.log
cat $OUT/summary.log
