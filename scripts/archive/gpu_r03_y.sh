#!/bin/bash
# inter-kernel gaps inside a replayed step graph: kernel trace of a short graph run
OUT=$PWD/gpurun_out/r03y
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o g -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other-form > $OUT/run.log 2>&1; echo "rc=$?" >> $OUT/summary.log
cd $GRAFT_REPO_ROOT
python scripts/gap_report.py gpurun_out/r03y/prof; exit 0
python - <<"PY"
import csv, glob, collections
f = glob.glob("gpurun_out/r03y/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last ~4 steps: take the last 2600 kernels
rows = rows[-2600:]
gaps = []
busy = 0
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps.append((g, a["Kernel_Name"][:60], b["Kernel_Name"][:60]))
    busy += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
pos = [g for g, _, _ in gaps if g > 0]
print(f"kernels {len(rows)} span {span/1e6:.2f} ms busy {busy/1e6:.2f} ms idle {(span-busy)/1e6:.2f} ms; positive gaps {len(pos)} mean {sum(pos)/max(1,len(pos)):.0f} ns; overlapping (negative) {sum(1 for g,_,_ in gaps if g<0)}")
hist = collections.Counter(min(g // 1000, 20) for g in pos)
print("gap histogram (us bucket: count):", sorted(hist.items()))
big = sorted(gaps, reverse=True)[:12]
for g, a, b in big: print(f"  {g/1e3:8.1f} us after {a} -> {b}")
PY
find $OUT/prof -name "*kernel_trace.csv" -delete
