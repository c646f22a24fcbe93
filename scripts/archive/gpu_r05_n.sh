#!/bin/bash
# round 5, call n: hand-issued K / V^T fragment reads in the multi-tile attention kernel (counted lgkmcnt, four ahead; no compiler-made vmcnt(0) drain in
# front of the transposing reads) against compiler-issued reads (libaa_mi355_syncfrags.so = -DAA_ATTN_SYNC_FRAGS=1): GPU tests, kernel and step A/B
OUT=gpurun_out/r05n; mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/animate_anything_amd/libaa_mi355_syncfrags.so
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "attention" > $OUT/tests_attention.log 2>&1; echo "attention tests rc=$?" >> $OUT/summary.log
for rep in 1 2; do
AA_LIBRARY=$V timeout 600 python scripts/bench_kernels.py --only "attn" --reps 20 > $OUT/attn_sync_$rep.log 2>&1
timeout 600 python scripts/bench_kernels.py --only "attn" --reps 20 > $OUT/attn_async_$rep.log 2>&1
done
for rep in 1 2 3; do
AA_LIBRARY=$V timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline > $OUT/bench_sync_$rep.json 2>$OUT/bench.err; echo "bench sync rc=$?" >> $OUT/summary.log
timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline > $OUT/bench_async_$rep.json 2>$OUT/bench.err; echo "bench async rc=$?" >> $OUT/summary.log
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "test_unet_forward_at_the_metric_configuration or three_steps or attention" > $OUT/tests.log 2>&1; echo "fullsize tests rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
tail -2 $OUT/tests_attention.log
for f in $OUT/attn_*.log; do echo $f; grep -iE "attn (spatial|cross)" $f; done
for f in $OUT/bench_sync_1.json $OUT/bench_async_1.json $OUT/bench_sync_2.json $OUT/bench_async_2.json $OUT/bench_sync_3.json $OUT/bench_async_3.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
tail -3 $OUT/tests.log
