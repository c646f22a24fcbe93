#!/bin/bash
OUT=gpurun_out/r03l
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -k "x_tiles_bf16" > $OUT/test_bf16.log 2>&1; echo "bf16 x tiles rc=$?" >> $OUT/summary.log
tail -2 $OUT/test_bf16.log
timeout 600 python scripts/bench_kernels.py --only "norm" > $OUT/bench_norm.log 2>&1
cat $OUT/summary.log; cat $OUT/bench_norm.log | grep -v amdgpu
