#!/bin/bash
# r04 call D: LayerNorm fold (row statistics from the producer's epilogue, correction in the consumer's): parity + bench
OUT=$PWD/gpurun_out/r04d
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_kernels.py -m gpu -q -x -n 4 -k "layernorm or upsample or tile_shapes or geglu" > $OUT/test_kernels.log 2>&1; echo "test_kernels rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_kernels.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_unet.py tests/test_gpu_config0.py -m gpu -q -x > $OUT/test_full.log 2>&1; echo "test_full rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_full.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-form --tile-cache $OUT/tile_cache.json --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
head -c 700 $OUT/bench.json; echo
tail -3 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-form --no-roofline --no-graph --tile-cache $OUT/tile_cache.json > $OUT/prof_run.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -25 $OUT/kernel_stats.csv | cut -c1-200
find $OUT/prof -name "*kernel_trace.csv" -delete
cat $OUT/summary.log
