#!/bin/bash
# round 5, call b: ff-out + proj_out as one two-source contraction (layers.FF_PROJ_MERGE) - A/B on one box, new signatures tuned first
OUT=gpurun_out/r05b
mkdir -p $OUT
export TMPDIR=/tmp
TC=$OUT/tile_cache.json
cp animate_anything_amd/tile_cache_gfx950.json $TC
timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-vae --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
for rep in 1 2; do
AA_FF_PROJ_MERGE=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --tile-cache $TC --gemm-breakdown $OUT/gemm_breakdown_off.txt > $OUT/bench_off_$rep.json 2>$OUT/bench_off.err; echo "bench off rc=$?" >> $OUT/summary.log
AA_FF_PROJ_MERGE=1 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --tile-cache $TC --gemm-breakdown $OUT/gemm_breakdown_on.txt > $OUT/bench_on_$rep.json 2>$OUT/bench_on.err; echo "bench on rc=$?" >> $OUT/summary.log
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "test_unet_forward_at_the_metric_configuration or three_steps" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
for f in $OUT/bench_off_1.json $OUT/bench_on_1.json $OUT/bench_off_2.json $OUT/bench_on_2.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['contraction_launches_per_step'], d['autotuned_signatures'])"; done
tail -4 $OUT/tests.log
head -30 $OUT/gemm_breakdown_on.txt
