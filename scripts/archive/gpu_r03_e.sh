#!/bin/bash
OUT=gpurun_out/r03e
mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -k "x_tiles or dma_tile_shapes or explicit_k_splits or geglu_forced" > $OUT/test_x.log 2>&1; echo "x tests rc=$?" >> $OUT/summary.log
tail -2 $OUT/test_x.log
timeout 1200 python -m pytest tests/test_gpu_config0.py tests/test_lora.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -s -k "config0 or gpu_fold or 16x48x48 or large_magnitude" > $OUT/test_new.log 2>&1; echo "new tests rc=$?" >> $OUT/summary.log
tail -60 $OUT/test_new.log
CF=3,12,15,27,33,36,40,41,43,44,45
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs $CF --only "linear" > $OUT/sweep_linear.log 2>&1; echo "sweep linear rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
python - <<'PY'
import re,collections
for f in ("gpurun_out/r03e/sweep_linear.log",):
    rows=collections.OrderedDict()
    for l in open(f):
        m=re.match(r"(.*?) \[(\d+):.*?\]\s+([\d.]+) us\s+([\d.]+) TFLOP",l)
        if m: rows.setdefault(m.group(1).strip(),{})[int(m.group(2))]=float(m.group(4))
    for k,v in rows.items():
        print(f"{k:48s} " + " ".join(f"{c}:{t:.0f}" for c,t in sorted(v.items())))
PY
