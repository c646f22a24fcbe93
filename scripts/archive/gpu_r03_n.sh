#!/bin/bash
# branch-free epilogue + bias folded into the accumulators: parity, phase probe, sweep
OUT=gpurun_out/r03n
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_kernels.py -m gpu -x -q > $OUT/test_kernels.log 2>&1; echo "kernel tests rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_kernels.log
PROBE_ONLY="geglu K=320" timeout 300 python scripts/phase_probe_x.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_probe_x.log
cut -c1-260 $OUT/phase_probe_x.log
CF=4,11,14,15,26,31,36,39,40,43,44,45
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs $CF --only "linear" > $OUT/sweep_linear.log 2>&1; echo "sweep linear rc=$?" >> $OUT/summary.log
timeout 900 python scripts/bench_kernels.py --cfg-sweep --cfgs 14,15,34,36,39,40,42 --only "conv" > $OUT/sweep_conv.log 2>&1; echo "sweep conv rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
python - <<'PY'
import re,collections
for f in ("gpurun_out/r03n/sweep_linear.log","gpurun_out/r03n/sweep_conv.log"):
    rows=collections.OrderedDict()
    for l in open(f):
        m=re.match(r"(.*?) \[(\d+):.*?\]\s+([\d.]+) us\s+([\d.]+) TFLOP",l)
        if m: rows.setdefault(m.group(1).strip(),{})[int(m.group(2))]=float(m.group(4))
    for k,v in rows.items():
        print(f"{k:48s} " + " ".join(f"{c}:{t:.0f}" for c,t in sorted(v.items())))
PY
