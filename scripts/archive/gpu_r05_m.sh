#!/bin/bash
# round 5, call m: attention ring of 2 slots + 4 workgroups per CU (A/B build, 7 spilled registers) against the library (3 slots, 3 per CU)
OUT=gpurun_out/r05m; mkdir -p $OUT
V=$PWD/animate_anything_amd/libaa_mi355_ring2.so
for rep in 1 2; do
AA_LIBRARY=$V timeout 600 python scripts/bench_kernels.py --only "attn spatial" --reps 20 > $OUT/attn_ring2_$rep.log 2>&1
timeout 600 python scripts/bench_kernels.py --only "attn spatial" --reps 20 > $OUT/attn_ring3_$rep.log 2>&1
done
for f in $OUT/attn_*.log; do echo $f; grep -i "attn" $f; done
