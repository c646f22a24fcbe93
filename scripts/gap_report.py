#!/usr/bin/env python
"""Idle time between the kernels of one replayed denoising step (rocprofv3 --kernel-trace csv): the kernels between two
consecutive cfg_dpm_step launches."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "cfg_dpm_step" in r["Kernel_Name"]]
print("kernels", len(rows), "cfg steps at", idx[-6:])
for a, b in zip(idx[-4:-1], idx[-3:]):
    seg = rows[a + 1:b + 1]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    gaps = [int(y["Start_Timestamp"]) - int(x["End_Timestamp"]) for x, y in zip(seg, seg[1:])]
    pos = [g for g in gaps if g > 0]
    hist = collections.Counter(min(g // 500, 12) for g in pos)
    print(f"step: {len(seg)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms, mean gap {sum(pos)/max(1,len(pos)):.0f} ns, "
          f"overlaps {sum(1 for g in gaps if g < 0)}; gap histogram (0.5 us buckets): {sorted(hist.items())}")
    worst = sorted(zip(gaps, [r['Kernel_Name'][:50] for r in seg], [r['Kernel_Name'][:50] for r in seg[1:]]), reverse=True)[:5]
    for g, x, y in worst:
        print(f"      {g/1e3:7.1f} us  {x} -> {y}")

# per-family totals of the last replayed step
import re
seg = rows[idx[-2] + 1:idx[-1] + 1]
fam = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r["Kernel_Name"]
    m = re.search(r"aa::(\w+)", n) or re.search(r"_ZN2aa\d+([a-z0-9_]+?)(?:I|E)", n)
    k = m.group(1) if m else n[:40]
    fam[k][0] += 1
    fam[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("last replayed step by kernel family (launches, ms):")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:32s} {v[0]:5d} {v[1] / 1e6:8.3f}")
print(f"  {'total':32s} {sum(v[0] for v in fam.values()):5d} {sum(v[1] for v in fam.values()) / 1e6:8.3f}")

# per kernel instance (name + template arguments + grid) of the last replayed step: launches, total, average
def short(n):
    n = re.sub(r"^_ZN2aa\d+", "", n)
    n = re.sub(r"void aa::", "", n)
    return n[:70]
inst = collections.defaultdict(lambda: [0, 0])
for r in seg:
    grid = "x".join(str(r.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if k in r) or str(r.get("Grid_Size", ""))
    k = (short(r["Kernel_Name"]), grid)
    inst[k][0] += 1
    inst[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("last replayed step by kernel instance and grid (launches, total ms, average us):")
for k, v in sorted(inst.items(), key=lambda kv: -kv[1][1])[:90]:
    print(f"  {k[0]:70s} grid {k[1]:>16s} {v[0]:4d} {v[1] / 1e6:8.3f} {v[1] / v[0] / 1e3:8.1f}")
