#!/bin/bash
OUT=gpurun_out/r03t
mkdir -p $OUT
CF=36,39,40,44,46,47
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs $CF --only "linear" > $OUT/sweep_linear_stagger.log 2>&1; echo "sweep stagger rc=$?" >> $OUT/summary.log
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs $CF --only "linear" --ablate 64 > $OUT/sweep_linear_nostagger.log 2>&1; echo "sweep nostagger rc=$?" >> $OUT/summary.log
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs 39,42 --only "conv" > $OUT/sweep_conv_stagger.log 2>&1
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs 39,42 --only "conv" --ablate 64 > $OUT/sweep_conv_nostagger.log 2>&1
cat $OUT/summary.log
python - <<'PY'
import re,collections
for f in ("sweep_linear_stagger","sweep_linear_nostagger","sweep_conv_stagger","sweep_conv_nostagger"):
    print(f)
    rows=collections.OrderedDict()
    for l in open(f"gpurun_out/r03t/{f}.log"):
        m=re.match(r"(.*?) \[(\d+):.*?\]\s+([\d.]+) us\s+([\d.]+) TFLOP",l)
        if m: rows.setdefault(m.group(1).strip(),{})[int(m.group(2))]=float(m.group(4))
    for k,v in rows.items():
        print(f"  {k:48s} " + " ".join(f"{c}:{t:.0f}" for c,t in sorted(v.items())))
PY
