#!/bin/bash
# round 3 closing run: tile cache for the three workloads, the whole GPU suite, bench lines, rocprof stats, PMC traffic
OUT=gpurun_out/r03z
mkdir -p $OUT
export TMPDIR=/tmp
TC=$OUT/tile_cache.json
timeout 1500 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --tile-cache $TC > $OUT/tune_unet3d.log 2>&1; echo "tune unet3d rc=$?" >> $OUT/summary.log
timeout 1500 python bench.py --workload svd --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --tile-cache $TC > $OUT/tune_svd.log 2>&1; echo "tune svd rc=$?" >> $OUT/summary.log
timeout 1500 python bench.py --workload rgba --steps 3 --warmup 2 --no-cpu-baseline --tile-cache $TC > $OUT/tune_rgba.log 2>&1; echo "tune rgba rc=$?" >> $OUT/summary.log
cp $TC animate_anything_amd/tile_cache_gfx950.json
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $OUT/summary.log
tail -3 $OUT/gpu_tests.log
timeout 900 python bench.py --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.json 2>$OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
timeout 900 python bench.py --workload svd --gemm-breakdown $OUT/svd_gemm_breakdown.txt > $OUT/bench_svd.json 2>$OUT/bench_svd.err; echo "bench svd rc=$?" >> $OUT/summary.log
timeout 900 python bench.py --workload rgba > $OUT/bench_rgba.json 2>$OUT/bench_rgba.err; echo "bench rgba rc=$?" >> $OUT/summary.log
timeout 900 python bench.py --dtype bf16 --no-cpu-baseline --no-other-form > $OUT/bench_bf16.json 2>/dev/null; echo "bench bf16 rc=$?" >> $OUT/summary.log
ROOT=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o p -- python $ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-other-form > $ROOT/$OUT/prof.log 2>&1; echo "rocprof rc=$?" >> $ROOT/$OUT/summary.log
cd $ROOT
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name "*kernel_trace.csv" -delete
bash scripts/pmc_traffic.sh r03z/traffic > $OUT/traffic.log 2>&1
cat $OUT/summary.log
cat $OUT/bench.json | cut -c1-1500
cat $OUT/bench_svd.json | cut -c1-700
cat $OUT/bench_rgba.json | cut -c1-700
head -25 $OUT/kernel_stats.csv | cut -c1-160
cat gpurun_out/r03z/traffic/traffic.json | head -40
