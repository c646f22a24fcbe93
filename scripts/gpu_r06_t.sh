#!/bin/bash
# round 6, call t: GroupNorm in front of proj_in applied inside aa_linear_rows (aa_groupnorm_coef, ABI 109): GPU parity, step A/B
# (default / AA_GN_FOLD=0 / AA_LINEAR_ROWS=0, three alternating rounds on one box), full-size parity, step trace
OUT=gpurun_out/r06t; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_linear_rows.py tests/test_ff_fused.py tests/test_seq_attention.py -m gpu -q -x --tb=short > $OUT/tests.log 2>&1; echo "kernel tests rc=$?" >> $OUT/summary.log
TC=$OUT/tile_cache.json
cp animate_anything_amd/tile_cache_gfx950.json $TC
timeout 1500 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
for rep in 1 2 3; do
AA_LINEAR_ROWS=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline > $OUT/ab_rows_off_$rep.json 2>$OUT/bench.err; echo "bench rows off rc=$?" >> $OUT/summary.log
AA_GN_FOLD=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/ab_gnfold_off_$rep.json 2>$OUT/bench.err; echo "bench gn fold off rc=$?" >> $OUT/summary.log
timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/ab_default_$rep.json 2>$OUT/bench.err; echo "bench default rc=$?" >> $OUT/summary.log
done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_unet.py -m gpu -q -x --tb=short -n 2 > $OUT/tests_fullsize.log 2>&1; echo "fullsize tests rc=$?" >> $OUT/summary.log
ROOT=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/gprof -o g -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-form --no-vae --tile-cache $ROOT/$TC > $ROOT/$OUT/gprof.log 2>&1; echo "graph trace rc=$?" >> $ROOT/$OUT/summary.log
cd $ROOT
python scripts/gap_report.py $OUT/gprof > $OUT/graph_step_kernels.txt 2>&1
find $OUT/gprof -name "*kernel_trace.csv" -delete
cat $OUT/summary.log; tail -3 $OUT/tests.log; tail -3 $OUT/tests_fullsize.log
for f in $OUT/ab_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
grep -A22 "by kernel family" $OUT/graph_step_kernels.txt
