#!/bin/bash
# r04 call I: LayerNorm statistics with 2 or 10 partial sums per row finalised in the consumer (independent loads), others through aa_ln_finalize
OUT=$PWD/gpurun_out/r04i
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_kernels.py -m gpu -q -x -n 4 -k "layernorm or k_split or splits" > $OUT/test_kernels.log 2>&1; echo "test_kernels rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_kernels.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_unet.py -m gpu -q -x -k "metric_configuration or small_unet or full_architecture" > $OUT/test_full.log 2>&1; echo "test_full rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_full.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-form --tile-cache $OUT/tile_cache.json --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
head -c 300 $OUT/bench.json; echo
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-form --no-roofline --tile-cache $OUT/tile_cache.json > $OUT/bench2.json 2> $OUT/bench2.err; echo "bench2 rc=$?" >> $OUT/summary.log
head -c 300 $OUT/bench2.json; echo
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-form --no-roofline --no-graph --tile-cache $OUT/tile_cache.json > $OUT/prof_run.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -8 $OUT/kernel_stats.csv | cut -c1-170
find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT/prof -name "*.db" -delete
cat $OUT/summary.log
