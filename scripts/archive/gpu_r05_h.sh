#!/bin/bash
# round 5, call h: GroupNorm statistics / apply kernels - rows in flight per thread (AA_GN_U_*), chunk targets (AA_GN_WANT / AA_GN_CAP),
# apply workgroups per statistics chunk (AA_GN_APPLY_MULT): two-kernel form of every norm shape of the step
OUT=gpurun_out/r05h
mkdir -p $OUT
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 300 python scripts/bench_kernels.py --only "groupnorm" --reps 20 > $OUT/gn_$tag.log 2>&1; echo "== $tag $@"; grep -E "two kernels|no one-kernel" $OUT/gn_$tag.log | grep -E "C=320 64x64|C=640 32x32|C=1280 16x16|C=640 64x64|C=960 64x64"; }
run base AA_X=0
run u8 AA_GN_U_STATS=8 AA_GN_U_APPLY=8
run u8s AA_GN_U_STATS=8
run u8a AA_GN_U_APPLY=8
run want2k AA_GN_WANT=2048
run want512 AA_GN_WANT=512
run cap512 AA_GN_CAP=512 AA_GN_WANT=2048
run cap128 AA_GN_CAP=128
run mult2 AA_GN_APPLY_MULT=2
run mult4 AA_GN_APPLY_MULT=4
run u8mult2 AA_GN_U_STATS=8 AA_GN_U_APPLY=8 AA_GN_APPLY_MULT=2
run base2 AA_X=0
