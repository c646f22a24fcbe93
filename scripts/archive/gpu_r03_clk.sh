#!/bin/bash
# sustained clock / power while the step graph replays
OUT=gpurun_out/r03clk
mkdir -p $OUT
( for i in $(seq 1 400); do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done > $OUT/smi.jsonl ) &
SMI=$!
timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-other-form > $OUT/bench.json 2>$OUT/bench.err
kill $SMI 2>/dev/null
python - <<'PY'
import json
rows=[]
for l in open("gpurun_out/r03clk/smi.jsonl"):
    l=l.strip()
    if not l.startswith("{"): continue
    try: d=json.loads(l)
    except Exception: continue
    c=d.get("card0",{})
    rows.append(c)
print(len(rows),"samples; keys:", list(rows[0].keys())[:12] if rows else None)
import re
def num(s):
    m=re.search(r"([\d.]+)", str(s)); return float(m.group(1)) if m else None
for key in (rows[0].keys() if rows else []):
    vals=[num(r.get(key)) for r in rows if num(r.get(key)) is not None]
    if vals and ("clk" in key.lower() or "power" in key.lower()):
        vals_s=sorted(vals); print(f"{key}: min {vals_s[0]} median {vals_s[len(vals_s)//2]} max {vals_s[-1]}")
PY
cut -c1-200 $OUT/bench.json
