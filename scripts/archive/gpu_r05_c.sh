#!/bin/bash
# round 5, call c: lazy row maximum in the attention kernel (AaAttention._pad bit 1 = the old eager form) - GPU tests, per-kernel A/B, step A/B
OUT=gpurun_out/r05c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "attention" > $OUT/tests_attention.log 2>&1; echo "attention tests rc=$?" >> $OUT/summary.log
for fl in 2 0 2 0; do
AA_ATTN_FLAGS=$fl timeout 600 python scripts/bench_kernels.py --only attn --reps 20 > $OUT/attn_flags_${fl}_$RANDOM.log 2>&1
done
for rep in 1 2; do
AA_ATTN_FLAGS=2 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline > $OUT/bench_eager_$rep.json 2>$OUT/bench.err; echo "bench eager rc=$?" >> $OUT/summary.log
AA_ATTN_FLAGS=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline > $OUT/bench_lazy_$rep.json 2>$OUT/bench.err; echo "bench lazy rc=$?" >> $OUT/summary.log
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "test_unet_forward_at_the_metric_configuration or three_steps or attention" > $OUT/tests.log 2>&1; echo "fullsize tests rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
tail -3 $OUT/tests_attention.log
for f in $OUT/attn_flags_*.log; do echo $f; grep -i "attn" $f; done
for f in $OUT/bench_eager_1.json $OUT/bench_lazy_1.json $OUT/bench_eager_2.json $OUT/bench_lazy_2.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'])"; done
tail -4 $OUT/tests.log
