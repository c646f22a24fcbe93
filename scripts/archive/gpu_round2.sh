#!/bin/bash
# One gpurun call: GPU parity suite, smoke, bench with fresh autotune (-> tile cache), bench on the cache, rocprofv3 kernel stats.
# Usage: scripts/gpu_round2.sh <tag> [skip-tests]
TAG=${1:-r02x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" != "skip-tests" ]; then
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $OUT/test_gpu.log 2>&1; echo "gpu suite rc=$?" >> $OUT/summary.log
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.log
fi
timeout 900 python bench.py --steps 20 --warmup 5 --no-tile-cache --tile-cache $OUT/tile_cache_gfx950.json --gemm-breakdown $OUT/gemm_breakdown.txt --no-cpu-baseline > $OUT/bench_autotune.log 2>&1; echo "bench(autotune) rc=$?" >> $OUT/summary.log
cp $OUT/tile_cache_gfx950.json animate_anything_amd/tile_cache_gfx950.json
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench(default, committed-cache path) rc=$?" >> $OUT/summary.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-other-form > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?" >> $OUT/summary.log
find $OUT/prof -type f ! -name "*stats*" -size +1M -delete
cat $OUT/summary.log
tail -4 $OUT/test_gpu.log
tail -1 $OUT/bench_autotune.log | cut -c1-400
tail -1 $OUT/bench.log | cut -c1-400
