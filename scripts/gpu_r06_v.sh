#!/bin/bash
# round 6, call v: aa_linear_rows with the epilogue through LDS (four lanes to a row) - GPU parity, isolation timing, step time on call t's tile choices
OUT=gpurun_out/r06v; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_linear_rows.py -m gpu -q -x --tb=short > $OUT/tests.log 2>&1; echo "kernel tests rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_linear_rows.py > $OUT/bench_linear_rows.txt 2>&1; echo "bench rc=$?" >> $OUT/summary.log
TC=$OUT/tile_cache.json
cp animate_anything_amd/tile_cache_gfx950.json $TC
timeout 1500 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
for rep in 1 2 3; do
AA_LINEAR_ROWS=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/ab_rows_off_$rep.json 2>$OUT/bench.err; echo "bench rows off rc=$?" >> $OUT/summary.log
timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/ab_default_$rep.json 2>$OUT/bench.err; echo "bench default rc=$?" >> $OUT/summary.log
done
cat $OUT/summary.log; tail -3 $OUT/tests.log; cat $OUT/bench_linear_rows.txt
for f in $OUT/ab_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
