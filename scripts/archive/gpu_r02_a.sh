#!/bin/bash
# Round-2 GPU call A: oracle pin probe, parity at the metric configuration, whole GPU suite, bench (+ tile cache), CPU baseline.
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{ python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.mem_get_info())"; nproc; free -g; lscpu | grep "Model name"; } > $OUT/env.log 2>&1
# (1) can the real reference be imported on the GPU box?  (diffusers==0.24.0 is the un-vendored dependency, requirements.txt:4)
{ for m in diffusers torchvision omegaconf cv2 imageio decord; do python -c "import $m; print('$m', $m.__version__)" 2>&1 | tail -1; done
  timeout 30 pip download diffusers==0.24.0 --no-deps -d /tmp/whl 2>&1 | tail -2
  ls /opt/wheelhouse 2>/dev/null | grep -i -E "diffusers|torchvision|omegaconf" ; echo "wheelhouse grep rc=$?"; } > $OUT/oracle_pin_probe.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short > $OUT/test_fullsize.log 2>&1; echo "fullsize rc=$?" >> $OUT/summary.log
timeout 900 python -m pytest tests -m gpu -q -x --tb=short --deselect tests/test_gpu_fullsize.py > $OUT/test_gpu.log 2>&1; echo "gpu suite rc=$?" >> $OUT/summary.log
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-tile-cache --tile-cache $OUT/tile_cache_gfx950.json --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/summary.log
timeout 600 python tests/golden/make_fullsize_golden.py --timing-json $OUT/cpu_baseline_fullsize.json > $OUT/cpu_fullsize.log 2>&1; echo "cpu fullsize rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
tail -c 1500 $OUT/test_fullsize.log
tail -3 $OUT/bench.log
