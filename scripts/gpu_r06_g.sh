#!/bin/bash
# round 6, call g: spatial self-attention with two 32-query blocks per wave (two waves per SIMD): parity, isolated timing, step A/B
OUT=gpurun_out/r06g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -k "attention" > $OUT/test_attn.log 2>&1; echo "attention tests rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_attention_qb.py > $OUT/bench_attn.log 2>&1; echo "bench_attn rc=$?" >> $OUT/summary.log
TC=$OUT/tile_cache.json
timeout 1500 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
for rep in 1 2; do
AA_ATTN_FLAGS=8 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_qb1_$rep.json 2>$OUT/bench.err; echo "bench qb1 rc=$?" >> $OUT/summary.log
AA_ATTN_FLAGS=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_qb2_$rep.json 2>$OUT/bench.err; echo "bench qb2 rc=$?" >> $OUT/summary.log
done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "metric_configuration or three_steps or attention" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
tail -3 $OUT/test_attn.log
grep images $OUT/bench_attn.log
for f in $OUT/bench_qb1_1.json $OUT/bench_qb2_1.json $OUT/bench_qb1_2.json $OUT/bench_qb2_2.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
tail -4 $OUT/tests.log
