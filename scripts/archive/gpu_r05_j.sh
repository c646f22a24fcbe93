#!/bin/bash
# round 5, call j: ABI 107 - LayerNorm coefficients written by a producer whose tile spans the row (no aa_ln_finalize launch): GPU tests,
# tile cache for the new version, A/B against the finalize launch (AA_PRODUCER_COEF=0), per-family trace
OUT=gpurun_out/r05j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "coefficients_from_the_producer or layernorm" > $OUT/tests_ln.log 2>&1; echo "ln tests rc=$?" >> $OUT/summary.log
TC=$OUT/tile_cache.json
timeout 1500 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --tile-cache $TC > $OUT/tune_unet3d.log 2>&1; echo "tune unet3d rc=$?" >> $OUT/summary.log
AA_PRODUCER_COEF=0 timeout 1500 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-vae --no-other-form --tile-cache $TC > $OUT/tune_unet3d_b.log 2>&1
for rep in 1 2 3; do
AA_PRODUCER_COEF=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_finalize_$rep.json 2>$OUT/bench.err; echo "bench finalize rc=$?" >> $OUT/summary.log
AA_PRODUCER_COEF=1 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_coef_$rep.json 2>$OUT/bench.err; echo "bench coef rc=$?" >> $OUT/summary.log
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "test_unet_forward_at_the_metric_configuration or three_steps" > $OUT/tests.log 2>&1; echo "fullsize tests rc=$?" >> $OUT/summary.log
ROOT=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/gprof -o g -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-form --no-vae --tile-cache $ROOT/$TC > $ROOT/$OUT/gprof.log 2>&1; echo "graph trace rc=$?" >> $ROOT/$OUT/summary.log
cd $ROOT
python scripts/gap_report.py $OUT/gprof > $OUT/graph_step_kernels.txt 2>&1
find $OUT/gprof -name "*kernel_trace.csv" -delete
cat $OUT/summary.log
tail -3 $OUT/tests_ln.log
for f in $OUT/bench_finalize_1.json $OUT/bench_coef_1.json $OUT/bench_finalize_2.json $OUT/bench_coef_2.json $OUT/bench_finalize_3.json $OUT/bench_coef_3.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
tail -3 $OUT/tests.log
grep -A16 "by kernel family" $OUT/graph_step_kernels.txt
