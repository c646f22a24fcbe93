#!/bin/bash
OUT=gpurun_out/r03o
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_kernels.py -m gpu -x -q > $OUT/test_kernels.log 2>&1; echo "kernel tests rc=$?" >> $OUT/summary.log
tail -2 $OUT/test_kernels.log
timeout 1500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-tile-cache --tile-cache $OUT/tile_cache.json --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.json 2>$OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
cut -c1-700 $OUT/bench.json
head -45 $OUT/gemm_breakdown.txt
