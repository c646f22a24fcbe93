#!/bin/bash
# Ablation of the contraction kernel: full / no operand DMA / no MFMA / neither, on a few shapes.
OUT=gpurun_out/${1:-abl}
mkdir -p $OUT
for ab in 0 1 2 3; do
  for only in "conv3x3 L320" "conv3x3 L1280 n" "linear ff2 L320" "linear qkv L320"; do
    python scripts/bench_kernels.py --only "$only" --cfg-sweep --ablate $ab 2>&1 | grep -E "\[(0|1|3|4):" | sed "s/^/ablate=$ab /" >> $OUT/ablate.log
  done
done
cat $OUT/ablate.log | sed -E 's/ +/ /g' | cut -c1-120
