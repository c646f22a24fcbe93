#!/bin/bash
# r04 call G: K splits that finish inside the kernel (tickets), new parity tests (qingming mask, bf16 full size, large-|q| 4096 x 5), bench
# (at the time of this call ops.USE_TICKETS defaulted to on and AA_NO_TICKETS=1 switched it off; after this measurement the default is off, AA_TICKETS=1 enables)
OUT=$PWD/gpurun_out/r04g
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_kernels.py -m gpu -q -x -n 4 > $OUT/test_kernels.log 2>&1; echo "test_kernels rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_kernels.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_unet.py -m gpu -q -x -s > $OUT/test_full.log 2>&1; echo "test_full rc=$?" >> $OUT/summary.log
grep -a "MSE\|max abs error\|passed\|failed" $OUT/test_full.log | tail -12
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-form --tile-cache $OUT/tile_cache.json --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
head -c 300 $OUT/bench.json; echo
AA_NO_TICKETS=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-form --no-roofline --tile-cache $OUT/tile_cache_notickets.json > $OUT/bench_notickets.json 2> $OUT/bench_notickets.err; echo "bench notickets rc=$?" >> $OUT/summary.log
head -c 300 $OUT/bench_notickets.json; echo
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-form --no-roofline --no-graph --tile-cache $OUT/tile_cache.json > $OUT/prof_run.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -16 $OUT/kernel_stats.csv | cut -c1-170
find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT/prof -name "*.db" -delete
cat $OUT/summary.log
