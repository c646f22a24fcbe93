#!/bin/bash
OUT=gpurun_out/r03s
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_kernels.py -m gpu -x -q > $OUT/test_kernels.log 2>&1; echo "kernel tests rc=$?" >> $OUT/summary.log
tail -2 $OUT/test_kernels.log
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs 36,39,40,43,44,45,46,47 --only "linear" > $OUT/sweep_linear.log 2>&1; echo "sweep linear rc=$?" >> $OUT/summary.log
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs 36,39,40,42,46,47 --only "conv" > $OUT/sweep_conv.log 2>&1; echo "sweep conv rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
python - <<'PY'
import re,collections
for f in ("gpurun_out/r03s/sweep_linear.log","gpurun_out/r03s/sweep_conv.log"):
    rows=collections.OrderedDict()
    for l in open(f):
        m=re.match(r"(.*?) \[(\d+):.*?\]\s+([\d.]+) us\s+([\d.]+) TFLOP",l)
        if m: rows.setdefault(m.group(1).strip(),{})[int(m.group(2))]=float(m.group(4))
    for k,v in rows.items():
        print(f"{k:48s} " + " ".join(f"{c}:{t:.0f}" for c,t in sorted(v.items())))
PY
