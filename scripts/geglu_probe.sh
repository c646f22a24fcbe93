#!/bin/bash
OUT=gpurun_out/${1:-geglu}; mkdir -p $OUT
for g in 160 64; do python scripts/bench_kernels.py --only "geglu" --cfg-sweep --geglu-gran $g 2>&1 | grep -v amdgpu | sed "s/^/gran=$g /" >> $OUT/geglu.log; done
sed -E 's/ +/ /g' $OUT/geglu.log | cut -c1-125
