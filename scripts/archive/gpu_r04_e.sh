#!/bin/bash
# r04 call E: LayerNorm fold with the rank-1 terms in the accumulator start; attention s_setprio A/B; per-kernel times of a step
OUT=$PWD/gpurun_out/r04f
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_kernels.py -m gpu -q -x -n 4 > $OUT/test_kernels.log 2>&1; echo "test_kernels rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_kernels.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_unet.py -m gpu -q -x > $OUT/test_full.log 2>&1; echo "test_full rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_full.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-form --tile-cache $OUT/tile_cache.json --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
head -c 400 $OUT/bench.json; echo
python scripts/bench_kernels.py --only "attn" > $OUT/attn_prio0.log 2>&1
AA_ATTN_FLAGS=1 python scripts/bench_kernels.py --only "attn" > $OUT/attn_prio1.log 2>&1
paste -d'\n' $OUT/attn_prio0.log $OUT/attn_prio1.log | cut -c1-150
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-form --no-roofline --no-graph --tile-cache $OUT/tile_cache.json > $OUT/prof_run.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -22 $OUT/kernel_stats.csv | cut -c1-180
find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT/prof -name "*.db" -delete
cat $OUT/summary.log
