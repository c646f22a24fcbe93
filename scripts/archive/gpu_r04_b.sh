#!/bin/bash
# r04 call B: parity of the gap-distributed im2col offsets (all kernel tests on the GPU), conv / tconv sweeps, one bench
OUT=$PWD/gpurun_out/r04b
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_kernels.py -m gpu -q -x -n 4 > $OUT/test_kernels.log 2>&1; echo "test_kernels rc=$?" >> $OUT/summary.log
tail -5 $OUT/test_kernels.log
timeout 600 python scripts/bench_kernels.py --only "conv" --cfg-sweep --cfgs 36,38,39,41,42,46,49 > $OUT/sweep_conv.log 2>&1; echo "sweep rc=$?" >> $OUT/summary.log
grep -E "auto|TFLOP" $OUT/sweep_conv.log | awk '{print}' | head -120
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-form --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
cat $OUT/bench.json | head -c 1500; cat $OUT/summary.log
