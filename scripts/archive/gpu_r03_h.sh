#!/bin/bash
OUT=gpurun_out/r03h
mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -k "x_tiles or dma_tile_shapes or explicit_k_splits or geglu_forced or persistent" > $OUT/test_x.log 2>&1; echo "x tests rc=$?" >> $OUT/summary.log
tail -2 $OUT/test_x.log
CF=14,15,26,39,40,44,46,47,48
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs $CF --only "linear" > $OUT/sweep_linear.log 2>&1; echo "sweep linear rc=$?" >> $OUT/summary.log
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs $CF --only "conv" > $OUT/sweep_conv.log 2>&1; echo "sweep conv rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
python - <<'PY'
import re,collections
for f in ("gpurun_out/r03h/sweep_linear.log","gpurun_out/r03h/sweep_conv.log"):
    rows=collections.OrderedDict()
    for l in open(f):
        m=re.match(r"(.*?) \[(\d+):.*?\]\s+([\d.]+) us\s+([\d.]+) TFLOP",l)
        if m: rows.setdefault(m.group(1).strip(),{})[int(m.group(2))]=float(m.group(4))
    for k,v in rows.items():
        print(f"{k:50s} " + " ".join(f"{c}:{t:.0f}" for c,t in sorted(v.items())))
PY
