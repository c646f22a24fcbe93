#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench (+ rocprof summary).  Usage: scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.mem_get_info())" > $OUT/env.log 2>&1
nproc >> $OUT/env.log; free -g >> $OUT/env.log
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short > $OUT/test_kernels.log 2>&1; echo "kernels rc=$?" >> $OUT/summary.log
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_pipeline.py -m gpu -q --tb=short > $OUT/test_unet.log 2>&1; echo "unet+pipeline rc=$?" >> $OUT/summary.log
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.log
timeout 900 python bench.py --steps 5 --warmup 2 --tile-cache $OUT/tile_cache.json --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/summary.log
timeout 900 python scripts/bench_kernels.py --cfg-sweep > $OUT/bench_kernels.log 2>&1; echo "bench_kernels rc=$?" >> $OUT/summary.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --tile-cache $OUT/tile_cache.json > $OUT/bench_nograph.log 2>&1; echo "bench_nograph rc=$?" >> $OUT/summary.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --tile-cache $GRAFT_REPO_ROOT/$OUT/tile_cache.json > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?" >> $OUT/summary.log
find $OUT/prof -name "*kernel_stats*" | head -3 >> $OUT/summary.log
# keep only the small stats files from the profile (traces can be large)
find $OUT/prof -type f ! -name "*stats*" -size +1M -delete
cat $OUT/summary.log
tail -3 $OUT/bench.log
