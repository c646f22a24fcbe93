#!/bin/bash
OUT=gpurun_out/r03w
mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "attention" > $OUT/test_attn.log 2>&1; echo "attn tests rc=$?" >> $OUT/summary.log
tail -2 $OUT/test_attn.log
timeout 600 python scripts/bench_kernels.py --only "attn temporal" > $OUT/bench_attn.log 2>&1
grep -v amdgpu $OUT/bench_attn.log | cut -c1-150
cat $OUT/summary.log
