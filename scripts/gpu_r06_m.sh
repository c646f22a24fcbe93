#!/bin/bash
# round 6, call m: grouped tile order inside an XCD's range (conv_gemm_x): contraction tests, step A/B (AA_DEBUG_ABLATE=16 = row-major), per-shape breakdown,
# PMC traffic by shape with the grouped order
OUT=gpurun_out/r06m; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels.py tests/test_gpu_tile_fuzz.py -m gpu -q -x --tb=short -n 3 > $OUT/tests.log 2>&1; echo "kernel + tile fuzz tests rc=$?" >> $OUT/summary.log
TC=$OUT/tile_cache.json
cp animate_anything_amd/tile_cache_gfx950.json $TC
timeout 1500 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
for rep in 1 2 3; do
AA_DEBUG_ABLATE=16 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_rowmajor_$rep.json 2>$OUT/bench.err; echo "bench row-major rc=$?" >> $OUT/summary.log
AA_DEBUG_ABLATE=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_grouped_$rep.json 2>$OUT/bench.err; echo "bench grouped rc=$?" >> $OUT/summary.log
done
AA_DEBUG_ABLATE=16 timeout 900 python bench.py --no-cpu-baseline --no-vae --no-other-form --tile-cache $TC --gemm-breakdown $OUT/gemm_breakdown_rowmajor.txt > $OUT/bench_rm.json 2>$OUT/bench.err
AA_DEBUG_ABLATE=0 timeout 900 python bench.py --no-cpu-baseline --no-vae --no-other-form --tile-cache $TC --gemm-breakdown $OUT/gemm_breakdown_grouped.txt > $OUT/bench_g.json 2>$OUT/bench.err
cat $OUT/summary.log
tail -3 $OUT/tests.log
for f in $OUT/bench_rowmajor_*.json $OUT/bench_grouped_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
bash scripts/pmc_traffic.sh r06m/traffic > $OUT/traffic.log 2>&1
tail -3 $OUT/traffic.log
