#!/bin/bash
# round 6, call i: aa_ff_fused prototype (FeedForward + proj_out as one kernel at 320 channels): parity on the GPU, timing against the two contractions, ablations
OUT=gpurun_out/r06i; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ff_fused.py tests/test_seq_attention.py -m gpu -q -x --tb=short > $OUT/test_ff.log 2>&1; echo "ff + seq tests rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_ff_fused.py > $OUT/bench_ff.log 2>&1; echo "bench_ff rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_ff_fused.py --ablate > $OUT/ablate.log 2>&1; echo "ablate rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
tail -3 $OUT/test_ff.log
grep "rows=" $OUT/bench_ff.log
grep "rows=" $OUT/ablate.log
tail -5 $OUT/bench_ff.log
