#!/bin/bash
# round 4: the measurement half of the closing run alone (bench lines, rocprof stats, replayed-step trace, PMC traffic) - the closing
# run of the final library landed on a box that ran every kernel 7-14 % slower than the pool's usual
OUT=gpurun_out/r04y
mkdir -p $OUT
export TMPDIR=/tmp
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
bash scripts/pmc_traffic.sh r04y/traffic > $OUT/traffic.log 2>&1
rocm-smi --showclocks --showpower 2>/dev/null | head -20 > $OUT/smi.txt
cat $OUT/summary.log
for f in bench bench_svd bench_rgba bench_bf16; do head -c 260 $OUT/$f.json; echo; done
tail -14 $OUT/graph_step_kernels.txt
