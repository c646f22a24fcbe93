#!/bin/bash
# round 4 closing run: tile cache for the three workloads (fresh autotune), the whole GPU suite on those choices, smoke, bench lines,
# rocprof stats (eager) + kernel trace of replayed steps, PMC traffic
OUT=gpurun_out/r04zz
mkdir -p $OUT
export TMPDIR=/tmp
TC=$OUT/tile_cache.json
timeout 1500 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --tile-cache $TC > $OUT/tune_unet3d.log 2>&1; echo "tune unet3d rc=$?" >> $OUT/summary.log
timeout 1500 python bench.py --workload svd --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --tile-cache $TC > $OUT/tune_svd.log 2>&1; echo "tune svd rc=$?" >> $OUT/summary.log
timeout 1500 python bench.py --workload rgba --steps 3 --warmup 2 --no-cpu-baseline --tile-cache $TC > $OUT/tune_rgba.log 2>&1; echo "tune rgba rc=$?" >> $OUT/summary.log
cp $TC animate_anything_amd/tile_cache_gfx950.json
timeout 2400 python -m pytest tests -m gpu -x -q -n 3 > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $OUT/summary.log
tail -3 $OUT/gpu_tests.log
timeout 600 python scripts/debug/fuzz_tiles_fullsize.py 80 6 11 > $OUT/fuzz.log 2>&1; echo "tile fuzz rc=$? $(grep -c ' ok' $OUT/fuzz.log) ok $(grep -c FAIL $OUT/fuzz.log) fail" >> $OUT/summary.log
AA_TICKETS=1 timeout 600 python scripts/debug/fuzz_tiles_fullsize.py 60 4 12 > $OUT/fuzz_tickets.log 2>&1; echo "tile fuzz (tickets) rc=$? $(grep -c ' ok' $OUT/fuzz_tickets.log) ok $(grep -c FAIL $OUT/fuzz_tickets.log) fail" >> $OUT/summary.log
AA_LN_RAW=1 timeout 600 python scripts/debug/fuzz_tiles_fullsize.py 40 2 13 > $OUT/fuzz_lnraw.log 2>&1; echo "tile fuzz (raw LN statistics) rc=$? $(grep -c ' ok' $OUT/fuzz_lnraw.log) ok $(grep -c FAIL $OUT/fuzz_lnraw.log) fail" >> $OUT/summary.log
timeout 600 python scripts/debug/fuzz_tiles_fullsize.py 40 3 14 bf16 > $OUT/fuzz_bf16.log 2>&1; echo "tile fuzz (bf16) rc=$? $(grep -c ' ok' $OUT/fuzz_bf16.log) ok $(grep -c FAIL $OUT/fuzz_bf16.log) fail" >> $OUT/summary.log
timeout 900 python scripts/debug/fuzz_tiles_fullsize.py 40 3 15 svd > $OUT/fuzz_svd.log 2>&1; echo "tile fuzz (svd) rc=$? $(grep -c ' ok' $OUT/fuzz_svd.log) ok $(grep -c FAIL $OUT/fuzz_svd.log) fail" >> $OUT/summary.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.log
tail -1 $OUT/smoke.log
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
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/gprof -o g -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-form > $ROOT/$OUT/gprof.log 2>&1; echo "graph trace rc=$?" >> $ROOT/$OUT/summary.log
cd $ROOT
python scripts/gap_report.py $OUT/gprof > $OUT/graph_step_kernels.txt 2>&1
find $OUT/gprof -name "*kernel_trace.csv" -delete
bash scripts/pmc_traffic.sh r04zz/traffic > $OUT/traffic.log 2>&1
cat $OUT/summary.log
cat $OUT/bench.json | cut -c1-1800
cat $OUT/bench_svd.json | cut -c1-500
cat $OUT/bench_rgba.json | cut -c1-500
cat $OUT/bench_bf16.json | cut -c1-300
head -22 $OUT/kernel_stats.csv | cut -c1-160
cat gpurun_out/r04zz/traffic/traffic.json | head -60
