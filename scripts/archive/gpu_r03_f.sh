#!/bin/bash
OUT=gpurun_out/r03f
mkdir -p $OUT
timeout 600 python -m pytest tests/test_lora.py -m gpu -q --tb=short > $OUT/test_lora.log 2>&1; echo "lora rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_lora.log
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs 15,33,40,43,44,45 --only "linear geglu" > $OUT/sweep_geglu.log 2>&1; echo "sweep rc=$?" >> $OUT/summary.log
cat $OUT/summary.log; grep -E "\[" $OUT/sweep_geglu.log | cut -c1-120
