#!/bin/bash
# round 5, call d: the two guidance halves of the UNet call on two streams (unet3d.CLIP_STREAMS = 2; independent branches of the captured
# graph) against the batched call - new half-size signatures tuned first, then A/B twice, parity at the metric configuration
OUT=gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
TC=$OUT/tile_cache.json
cp animate_anything_amd/tile_cache_gfx950.json $TC
AA_CLIP_STREAMS=2 timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-vae --no-other-form --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
for rep in 1 2; do
AA_CLIP_STREAMS=1 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_s1_$rep.json 2>$OUT/bench_s1.err; echo "bench 1 stream rc=$?" >> $OUT/summary.log
AA_CLIP_STREAMS=2 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_s2_$rep.json 2>$OUT/bench_s2.err; echo "bench 2 streams rc=$?" >> $OUT/summary.log
done
AA_CLIP_STREAMS=2 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --no-graph --steps 5 --tile-cache $TC > $OUT/bench_s2_nograph.json 2>$OUT/bench_s2ng.err; echo "bench 2 streams eager rc=$?" >> $OUT/summary.log
AA_CLIP_STREAMS=2 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "test_unet_forward_at_the_metric_configuration or three_steps" > $OUT/tests.log 2>&1; echo "tests (2 streams) rc=$?" >> $OUT/summary.log
ROOT=$PWD
cd /tmp
AA_CLIP_STREAMS=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/gprof -o g -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-form --no-vae --tile-cache $ROOT/$TC > $ROOT/$OUT/gprof.log 2>&1; echo "graph trace rc=$?" >> $ROOT/$OUT/summary.log
cd $ROOT
python scripts/gap_report.py $OUT/gprof > $OUT/graph_step_kernels.txt 2>&1
find $OUT/gprof -name "*kernel_trace.csv" -delete
cat $OUT/summary.log
for f in $OUT/bench_s1_1.json $OUT/bench_s2_1.json $OUT/bench_s1_2.json $OUT/bench_s2_2.json $OUT/bench_s2_nograph.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
tail -5 $OUT/tests.log
head -12 $OUT/graph_step_kernels.txt
tail -5 $OUT/bench_s2.err
