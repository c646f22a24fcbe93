#!/bin/bash
# round 6, call e: is the fused temporal attention kernel's global traffic bound by the address coalescer (32-byte pieces per row)?
OUT=gpurun_out/r06e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/bench_seq_attention.py --ablate > $OUT/ablate.log 2>&1; echo "ablate rc=$?" >> $OUT/summary.log
cat $OUT/summary.log; cat $OUT/ablate.log
