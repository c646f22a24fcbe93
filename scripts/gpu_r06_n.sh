#!/bin/bash
# round 6, call n: aa_linear_rows - GPU parity + isolation timing against the tile family
OUT=gpurun_out/r06n; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_linear_rows.py -m gpu -q -x --tb=short > $OUT/tests.log 2>&1; echo "linear_rows tests rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_linear_rows.py > $OUT/bench_linear_rows.txt 2>&1; echo "bench rc=$?" >> $OUT/summary.log
cat $OUT/summary.log; tail -3 $OUT/tests.log; cat $OUT/bench_linear_rows.txt
