#!/bin/bash
# round 5, call o: self-attention Q|K|V projections packed head-major (ABI 108 AaAttnOperand.head_stride) against [Q | K | V] (AA_QKV_HEAD_MAJOR=0)
OUT=gpurun_out/r05o; mkdir -p $OUT
export TMPDIR=/tmp
TC=$OUT/tile_cache.json
timeout 1500 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
for rep in 1 2 3; do
AA_QKV_HEAD_MAJOR=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_qkv_$rep.json 2>$OUT/bench.err; echo "bench qkv rc=$?" >> $OUT/summary.log
AA_QKV_HEAD_MAJOR=1 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_headmajor_$rep.json 2>$OUT/bench.err; echo "bench head-major rc=$?" >> $OUT/summary.log
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_unet.py -x -q -s -k "test_unet_forward_at_the_metric_configuration or three_steps or small or attention" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/summary.log
ROOT=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/gprof -o g -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-form --no-vae --tile-cache $ROOT/$TC > $ROOT/$OUT/gprof.log 2>&1; echo "graph trace rc=$?" >> $ROOT/$OUT/summary.log
cd $ROOT
python scripts/gap_report.py $OUT/gprof > $OUT/graph_step_kernels.txt 2>&1
find $OUT/gprof -name "*kernel_trace.csv" -delete
cat $OUT/summary.log
for f in $OUT/bench_qkv_1.json $OUT/bench_headmajor_1.json $OUT/bench_qkv_2.json $OUT/bench_headmajor_2.json $OUT/bench_qkv_3.json $OUT/bench_headmajor_3.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
tail -3 $OUT/tests.log
grep -A16 "by kernel family" $OUT/graph_step_kernels.txt
grep "attention" $OUT/graph_step_kernels.txt | tail -14
