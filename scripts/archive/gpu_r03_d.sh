#!/bin/bash
OUT=gpurun_out/r03d
mkdir -p $OUT
timeout 1800 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --tile-cache $OUT/tile_cache.json --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/summary.log
cat $OUT/summary.log; tail -2 $OUT/bench.log | cut -c1-600; cat $OUT/gemm_breakdown.txt
