#!/bin/bash
# round 6, call o: aa_linear_rows with the last round split over its stages - GPU parity, isolation timing, step A/B (AA_LINEAR_ROWS=0 / 1), step trace
OUT=gpurun_out/r06o; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_linear_rows.py tests/test_ff_fused.py -m gpu -q -x --tb=short > $OUT/tests.log 2>&1; echo "linear_rows tests rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_linear_rows.py > $OUT/bench_linear_rows.txt 2>&1; echo "bench rc=$?" >> $OUT/summary.log
AA_LINEAR_ROWS_DEBUG=2 timeout 600 python scripts/bench_linear_rows.py > $OUT/bench_linear_rows_nosplit.txt 2>&1
TC=$OUT/tile_cache.json
cp animate_anything_amd/tile_cache_gfx950.json $TC
timeout 1500 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
for rep in 1 2 3; do
AA_LINEAR_ROWS=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline > $OUT/ab_off_$rep.json 2>$OUT/bench.err; echo "bench off rc=$?" >> $OUT/summary.log
AA_LINEAR_ROWS=1 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/ab_on_$rep.json 2>$OUT/bench.err; echo "bench on rc=$?" >> $OUT/summary.log
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x --tb=short -k "unet or pipeline or step" > $OUT/tests_fullsize.log 2>&1; echo "fullsize tests rc=$?" >> $OUT/summary.log
ROOT=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/gprof -o g -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-form --no-vae --tile-cache $ROOT/$TC > $ROOT/$OUT/gprof.log 2>&1; echo "graph trace rc=$?" >> $ROOT/$OUT/summary.log
cd $ROOT
python scripts/gap_report.py $OUT/gprof > $OUT/graph_step_kernels.txt 2>&1
find $OUT/gprof -name "*kernel_trace.csv" -delete
cat $OUT/summary.log; tail -3 $OUT/tests.log; tail -3 $OUT/tests_fullsize.log; cat $OUT/bench_linear_rows.txt; echo "--- no split"; cat $OUT/bench_linear_rows_nosplit.txt
for f in $OUT/ab_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
grep -A22 "by kernel family" $OUT/graph_step_kernels.txt
