#!/bin/bash
# round 5, call k: GroupNorm chunk count per image group - workgroups per launch just below a multiple of the 256 CUs (AA_GN_WANT experiment knob)
OUT=gpurun_out/r05k
mkdir -p $OUT
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 300 python scripts/bench_kernels.py --only "groupnorm" --reps 30 > $OUT/gn_$tag.log 2>&1; echo "== $tag $@"; grep -E "two kernels|no one-kernel" $OUT/gn_$tag.log | grep -E "2d.*(C=320 64x64|C=640 32x32|C=1280 16x16|C=640 64x64|C=960 64x64|C=1280 32x32|C=1920 32x32)"; }
run w1054 AA_GN_WANT=1054
run w1024 AA_GN_WANT=1024
run w768 AA_GN_WANT=768
run w1280 AA_GN_WANT=1280
run w1536 AA_GN_WANT=1536
run w2048 AA_GN_WANT=2048
run w512 AA_GN_WANT=512
run w1054b AA_GN_WANT=1054
run w1024b AA_GN_WANT=1024
