#!/bin/bash
OUT=gpurun_out/r03v
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_kernels.py -m gpu -x -q > $OUT/test_kernels.log 2>&1; echo "kernel tests rc=$?" >> $OUT/summary.log
tail -2 $OUT/test_kernels.log
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs 36,39,40,44 --only "L320" > $OUT/sweep_l0.log 2>&1; echo "sweep rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
grep -v amdgpu $OUT/sweep_l0.log | cut -c1-120
