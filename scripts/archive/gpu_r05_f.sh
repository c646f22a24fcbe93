#!/bin/bash
# round 5, call f: attention without the spills of the first lazy-maximum build; short-key-sequence kernel (text cross-attention) against
# the general kernel (AaAttention._pad bit 2); GPU tests of both; step + per-instance trace
OUT=gpurun_out/r05f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "attention" > $OUT/tests_attention.log 2>&1; echo "attention tests rc=$?" >> $OUT/summary.log
for fl in 4 0 4 0; do
AA_ATTN_FLAGS=$fl timeout 600 python scripts/bench_kernels.py --only attn --reps 20 > $OUT/attn_flags_${fl}_$RANDOM.log 2>&1
done
TC=$OUT/tile_cache.json
cp animate_anything_amd/tile_cache_gfx950.json $TC
timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-vae --no-other-form --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
for rep in 1 2; do
AA_ATTN_FLAGS=4 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_general_$rep.json 2>$OUT/bench.err; echo "bench general rc=$?" >> $OUT/summary.log
AA_ATTN_FLAGS=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_shortkv_$rep.json 2>$OUT/bench.err; echo "bench shortkv rc=$?" >> $OUT/summary.log
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "test_unet_forward_at_the_metric_configuration or three_steps or attention" > $OUT/tests.log 2>&1; echo "fullsize tests rc=$?" >> $OUT/summary.log
ROOT=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/gprof -o g -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-form --no-vae --tile-cache $ROOT/$TC > $ROOT/$OUT/gprof.log 2>&1; echo "graph trace rc=$?" >> $ROOT/$OUT/summary.log
cd $ROOT
python scripts/gap_report.py $OUT/gprof > $OUT/graph_step_kernels.txt 2>&1
find $OUT/gprof -name "*kernel_trace.csv" -delete
cat $OUT/summary.log
tail -3 $OUT/tests_attention.log
for f in $OUT/attn_flags_*.log; do echo $f; grep -i "attn" $f; done
for f in $OUT/bench_general_1.json $OUT/bench_shortkv_1.json $OUT/bench_general_2.json $OUT/bench_shortkv_2.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'])"; done
tail -4 $OUT/tests.log
grep -A16 "by kernel family" $OUT/graph_step_kernels.txt
grep "attention" $OUT/graph_step_kernels.txt | tail -14
