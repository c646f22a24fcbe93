#!/bin/bash
OUT=gpurun_out/r03c
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -k "x_tiles or dma_tile_shapes or explicit_k_splits or geglu_forced" > $OUT/test_x.log 2>&1; echo "x tests rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_x.log
CF=3,4,14,15,26,27,34,35,36,37,38,39,40,41,42,43
timeout 1200 python scripts/bench_kernels.py --cfg-sweep --cfgs $CF --only "linear" > $OUT/sweep_linear.log 2>&1; echo "sweep linear rc=$?" >> $OUT/summary.log
timeout 1200 python scripts/bench_kernels.py --cfg-sweep --cfgs $CF --only "conv" > $OUT/sweep_conv.log 2>&1; echo "sweep conv rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
python - <<'PY'
import re,collections
for f in ("gpurun_out/r03c/sweep_linear.log","gpurun_out/r03c/sweep_conv.log"):
    rows=collections.OrderedDict()
    for l in open(f):
        m=re.match(r"(.*?) \[(\d+):.*?\]\s+([\d.]+) us\s+([\d.]+) TFLOP",l)
        if m: rows.setdefault(m.group(1).strip(),{})[int(m.group(2))]=float(m.group(4))
    for k,v in rows.items():
        best_old=max((t,c) for c,t in v.items() if c<36); best_x=max((t,c) for c,t in v.items() if c>=36)
        print(f"{k:52s} old {best_old[0]:7.1f} [{best_old[1]:2d}]  x {best_x[0]:7.1f} [{best_x[1]:2d}]  " + " ".join(f"{c}:{t:.0f}" for c,t in sorted(v.items()) if c>=36))
PY
