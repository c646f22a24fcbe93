#!/bin/bash
OUT=gpurun_out/r03k
mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x --tb=short -k "attention" > $OUT/test_attn.log 2>&1; echo "attn tests rc=$?" >> $OUT/summary.log
tail -2 $OUT/test_attn.log
timeout 600 python scripts/bench_kernels.py --only "attn" > $OUT/bench_attn.log 2>&1
cat $OUT/summary.log; cat $OUT/bench_attn.log | grep -v amdgpu
