#!/bin/bash
# round 5, call a: the round-4 kernels + this round's correctness work on the GPU - 3-step oracle parity of both guidance forms with a
# measured noise floor, LayerNorm-fold cancellation / split-off-tail tests, tile fuzz variants, the bench line with the `vae` object
# (VAE signatures join the tile cache), a kernel trace of replayed steps as the round's baseline.
OUT=gpurun_out/r05a
mkdir -p $OUT
export TMPDIR=/tmp
TC=$OUT/tile_cache.json
cp animate_anything_amd/tile_cache_gfx950.json $TC
timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-other-form --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
cp $TC animate_anything_amd/tile_cache_gfx950.json
AA_PARITY_REPORT=$OUT/parity_3steps.txt timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "three_steps or shared_prefix" > $OUT/tests_parity.log 2>&1; echo "parity tests rc=$?" >> $OUT/summary.log
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "cancellation or split_off_last_round or layernorm_fold" > $OUT/tests_ln.log 2>&1; echo "ln tests rc=$?" >> $OUT/summary.log
timeout 1500 python -m pytest tests/test_gpu_tile_fuzz.py -x -q -s > $OUT/tests_fuzz.log 2>&1; echo "fuzz tests rc=$?" >> $OUT/summary.log
timeout 900 python bench.py --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.json 2>$OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
ROOT=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/gprof -o g -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-form --no-vae > $ROOT/$OUT/gprof.log 2>&1; echo "graph trace rc=$?" >> $ROOT/$OUT/summary.log
cd $ROOT
python scripts/gap_report.py $OUT/gprof > $OUT/graph_step_kernels.txt 2>&1
find $OUT/gprof -name "*kernel_trace.csv" -delete
cat $OUT/summary.log
tail -5 $OUT/tests_parity.log; cat $OUT/parity_3steps.txt
tail -3 $OUT/tests_ln.log; tail -8 $OUT/tests_fuzz.log
cut -c1-2500 $OUT/bench.json
tail -14 $OUT/graph_step_kernels.txt
