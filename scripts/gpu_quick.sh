#!/bin/bash
# Quick perf iteration: GPU kernel parity tests + bench with the per-shape contraction breakdown.
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short > $OUT/test_kernels.log 2>&1; echo "kernels rc=$?" >> $OUT/summary.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/summary.log
[ -n "$2" ] && timeout 600 python scripts/bench_kernels.py --only "$2" --cfg-sweep > $OUT/bench_kernels.log 2>&1
cat $OUT/summary.log; tail -1 $OUT/bench.log | cut -c1-300; cat $OUT/gemm_breakdown.txt
