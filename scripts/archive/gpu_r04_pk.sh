#!/bin/bash
# r04: the LayerNorm fold's accumulator start and rstd scale as packed fp32 pairs: phase probe, bench, tile fuzz, the whole GPU suite
OUT=gpurun_out/r04pk
mkdir -p $OUT
PROBE_ONLY=none timeout 300 python scripts/phase_probe_ln.py > $OUT/phase_ln.log 2>&1; grep -a "geglu\|qkv" $OUT/phase_ln.log | cut -c1-250
timeout 600 python bench.py --no-cpu-baseline --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.json 2>$OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
head -c 250 $OUT/bench.json | cut -c60-250; echo
timeout 400 python scripts/debug/fuzz_tiles_fullsize.py 40 3 61 > $OUT/fuzz.log 2>&1; echo "tile fuzz rc=$? $(grep -c ' ok' $OUT/fuzz.log) ok $(grep -c FAIL $OUT/fuzz.log) fail" >> $OUT/summary.log
timeout 1200 python -m pytest tests -m gpu -q -n 3 > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $OUT/summary.log
tail -2 $OUT/gpu_tests.log
cat $OUT/summary.log
