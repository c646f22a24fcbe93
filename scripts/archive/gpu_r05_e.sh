#!/bin/bash
# round 5, call e: conv_out through the tiled path (8 padded output channels); per-instance kernel report of a replayed step
OUT=gpurun_out/r05e
mkdir -p $OUT
export TMPDIR=/tmp
TC=$OUT/tile_cache.json
cp animate_anything_amd/tile_cache_gfx950.json $TC
timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-vae --no-other-form --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
timeout 600 python bench.py --no-cpu-baseline --no-vae --tile-cache $TC --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.json 2>$OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "test_unet_forward_at_the_metric_configuration or three_steps" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/summary.log
ROOT=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/gprof -o g -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-form --no-vae --tile-cache $ROOT/$TC > $ROOT/$OUT/gprof.log 2>&1; echo "graph trace rc=$?" >> $ROOT/$OUT/summary.log
cd $ROOT
python scripts/gap_report.py $OUT/gprof > $OUT/graph_step_kernels.txt 2>&1
find $OUT/gprof -name "*kernel_trace.csv" -delete
cat $OUT/summary.log
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['autotuned_signatures'], d['other_form']['ms_per_step'])"
tail -3 $OUT/tests.log
tail -110 $OUT/graph_step_kernels.txt
