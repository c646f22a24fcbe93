#!/bin/bash
# round 6, call k: kernel trace of replayed steps with aa_seq_self_attention + aa_ff_fused in the step (where the step's time is now)
OUT=gpurun_out/r06k; mkdir -p $OUT
export TMPDIR=/tmp
TC=$OUT/tile_cache.json
cp animate_anything_amd/tile_cache_gfx950.json $TC
timeout 1500 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
ROOT=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/gprof -o g -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-form --no-vae --tile-cache $ROOT/$TC > $ROOT/$OUT/gprof.log 2>&1; echo "graph trace rc=$?" >> $ROOT/$OUT/summary.log
cd $ROOT
python scripts/gap_report.py $OUT/gprof > $OUT/graph_step_kernels.txt 2>&1
find $OUT/gprof -name "*kernel_trace.csv" -delete
timeout 900 python bench.py --no-cpu-baseline --no-vae --no-other-form --tile-cache $TC --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.json 2>$OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
grep -A70 "by kernel family" $OUT/graph_step_kernels.txt
