#!/bin/bash
# round 6, call j: aa_ff_fused wired into the transformers at 320 channels: parity (kernel tests, full-size UNet goldens), isolated timing, step A/B
# (AA_FF_FUSED=0 | 1, AA_FF_SPLIT=0 | 1)
OUT=gpurun_out/r06j; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ff_fused.py tests/test_seq_attention.py -m gpu -q -x --tb=short > $OUT/test_ff.log 2>&1; echo "ff + seq tests rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_ff_fused.py > $OUT/bench_ff.log 2>&1; echo "bench_ff rc=$?" >> $OUT/summary.log
TC=$OUT/tile_cache.json
AA_FF_FUSED=0 timeout 1500 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
AA_FF_FUSED=1 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune2.log 2>&1
for rep in 1 2 3; do
AA_FF_FUSED=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_off_$rep.json 2>$OUT/bench.err; echo "bench off rc=$?" >> $OUT/summary.log
AA_FF_FUSED=1 AA_FF_SPLIT=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_on_nosplit_$rep.json 2>$OUT/bench.err; echo "bench on (no split) rc=$?" >> $OUT/summary.log
AA_FF_FUSED=1 AA_FF_SPLIT=1 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_on_split_$rep.json 2>$OUT/bench.err; echo "bench on (split) rc=$?" >> $OUT/summary.log
done
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "metric_configuration or three_steps" > $OUT/tests.log 2>&1; echo "fullsize tests rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
tail -3 $OUT/test_ff.log
grep "rows=" $OUT/bench_ff.log
for f in $OUT/bench_off_*.json $OUT/bench_on_nosplit_*.json $OUT/bench_on_split_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
tail -4 $OUT/tests.log
