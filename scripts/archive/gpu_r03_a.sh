#!/bin/bash
# round 3, call A: the hand-scheduled tiles on hardware - parity first, then every contraction shape under every tile
OUT=gpurun_out/r03a
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -k "x_tiles or dma_tile_shapes or explicit_k_splits or geglu_forced" > $OUT/test_x.log 2>&1; echo "x tests rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_x.log
timeout 1200 python scripts/bench_kernels.py --cfg-sweep --only "linear" > $OUT/sweep_linear.log 2>&1; echo "sweep linear rc=$?" >> $OUT/summary.log
timeout 1200 python scripts/bench_kernels.py --cfg-sweep --only "conv" > $OUT/sweep_conv.log 2>&1; echo "sweep conv rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
grep -E "auto|\[(4|14|26|34|35|3|15|27|36|37|38|39|40):" $OUT/sweep_linear.log $OUT/sweep_conv.log | cut -c1-150
