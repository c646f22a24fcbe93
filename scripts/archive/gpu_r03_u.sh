#!/bin/bash
OUT=gpurun_out/r03u
mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "49 or x_tiles" > $OUT/test_kernels.log 2>&1; echo "kernel tests rc=$?" >> $OUT/summary.log
tail -2 $OUT/test_kernels.log
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs 39,42,47,49 --only "L320" > $OUT/sweep_l0.log 2>&1; echo "sweep rc=$?" >> $OUT/summary.log
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs 39,42,47,49 --only "L0" > $OUT/sweep_l0b.log 2>&1
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs 39,42,47,49 --only "L640" > $OUT/sweep_l1.log 2>&1
cat $OUT/summary.log
python - <<'PY'
import re,collections
for f in ("sweep_l0","sweep_l0b","sweep_l1"):
    rows=collections.OrderedDict()
    for l in open(f"gpurun_out/r03u/{f}.log"):
        m=re.match(r"(.*?) \[(\d+|auto).*?\]\s+([\d.]+) us\s+([\d.]+) TFLOP",l)
        if m: rows.setdefault(m.group(1).strip(),{})[m.group(2)]=(float(m.group(3)),float(m.group(4)))
    for k,v in rows.items():
        print(f"  {k:48s} " + " ".join(f"{c}:{t[1]:.0f}({t[0]:.0f}us)" for c,t in sorted(v.items())))
PY
