#!/usr/bin/env python
"""Reduce the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_traffic.sh to per-kernel-family HBM bytes per launch.
Units: rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE counts 128-byte requests of wide (16 B/lane)
coalesced reads at 64 B, i.e. HALF the bytes (MI355X_MICROARCH.md, HBM) -> fetch bytes are doubled here (upper
bound for narrow accesses).  Two or three identical eager steps are profiled (warm-up, timed, launch list): per-launch means are step-independent."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]


def family(name):
    if "conv_gemm_dma_kernel" in name or "conv3x3_slab_kernel" in name or "conv_gemm_x_kernel" in name:
        return "contraction_kernels"                       # the LDS-DMA implicit-GEMM family (compiled, halo-slab and hand-scheduled tiles)
    for key in ("ff_fused_kernel", "linear_rows_kernel", "groupnorm_coef", "conv_gemm_kernel", "splitk_reduce", "attention_kernel", "attention_shortkv", "groupnorm_apply",
                "groupnorm_stats", "groupnorm_fused", "layernorm_kernel", "ln_finalize", "cfg_dpm_step"):
        if key == "attention_kernel" and "seq_self_attention_kernel" in name:
            return "seq_self_attention_kernel"
        if key in name:
            return key
    return None


def collect(counter):
    acc = defaultdict(list)
    for f in glob.glob(f"{root}/{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if fam and r["Counter_Name"] == counter:
                acc[fam].append(float(r["Counter_Value"]))
    return acc


def by_shape(launch_list_path):
    """Per contraction SHAPE: HBM bytes measured (per-dispatch PMC rows of the contraction family in dispatch order, the last step of the
    pass) against the algorithmic bytes of the calls (bench.py --launch-list: the same step's aa_conv_gemm calls in launch order)."""
    calls = json.load(open(launch_list_path))
    n = sum(c["launches"] for c in calls)

    def ordered(counter):
        rows = []
        for f in glob.glob(f"{root}/{counter}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if family(r["Kernel_Name"]) == "contraction_kernels" and r["Counter_Name"] == counter:
                    rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        rows.sort()
        return [v for _, v in rows]

    f, w = ordered("FETCH_SIZE"), ordered("WRITE_SIZE")
    assert len(f) >= n and len(w) >= n and len(f) % n == 0 and len(w) % n == 0, (len(f), len(w), n)
    f, w = f[-n:], w[-n:]
    groups, i = {}, 0
    for c in calls:
        key = (c["M"], c["K"], c["N"], c["kind"], c["geglu"], c["two_source"], c["residual"])
        g = groups.setdefault(key, {"calls": 0, "launches": 0, "algorithmic_bytes": 0, "fetch_bytes": 0.0, "write_bytes": 0.0, "tile": c["tile"]})
        g["calls"] += 1
        g["launches"] += c["launches"]
        g["algorithmic_bytes"] += c["algorithmic_bytes"]
        for _ in range(c["launches"]):
            g["fetch_bytes"] += 2.0 * 1024.0 * f[i]
            g["write_bytes"] += 1024.0 * w[i]
            i += 1
    out = []
    for key, g in groups.items():
        meas = g["fetch_bytes"] + g["write_bytes"]
        out.append({"M": key[0], "K": key[1], "N": key[2], "kind": key[3], "geglu": key[4], "two_source": key[5], "residual": key[6], "tile": g["tile"],
                    "calls_per_step": g["calls"], "launches_per_step": g["launches"],
                    "algorithmic_MB_per_step": round(g["algorithmic_bytes"] / 1e6, 1), "measured_MB_per_step": round(meas / 1e6, 1),
                    "fetch_MB_per_step": round(g["fetch_bytes"] / 1e6, 1), "write_MB_per_step": round(g["write_bytes"] / 1e6, 1),
                    "measured_over_algorithmic": round(meas / g["algorithmic_bytes"], 3), "excess_MB_per_step": round((meas - g["algorithmic_bytes"]) / 1e6, 1)})
    out.sort(key=lambda r: -r["excess_MB_per_step"])
    tot_a, tot_m = sum(r["algorithmic_MB_per_step"] for r in out), sum(r["measured_MB_per_step"] for r in out)
    return {"shapes": out, "total_algorithmic_MB_per_step": round(tot_a, 1), "total_measured_MB_per_step": round(tot_m, 1),
            "total_measured_over_algorithmic": round(tot_m / tot_a, 3)}


if "--by-shape" in sys.argv:
    import os
    res = by_shape(sys.argv[sys.argv.index("--by-shape") + 1])
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from animate_anything_amd import build  # noqa: E402
    res["library_source_sha256_16"] = build.source_id()
    print(json.dumps(res, indent=1))
    sys.exit(0)

import os
STEPS = 3 if os.path.exists(f"{root}/launch_list_FETCH_SIZE.json") else 2      # warm-up + timed (+ the launch-list step of scripts/pmc_traffic.sh)
fetch, write = collect("FETCH_SIZE"), collect("WRITE_SIZE")
out = {}
for fam in sorted(set(fetch) | set(write)):
    f, w = fetch.get(fam, []), write.get(fam, [])
    fb = 2.0 * 1024.0 * sum(f) / max(len(f), 1)
    wb = 1024.0 * sum(w) / max(len(w), 1)
    out[fam] = {"launches_profiled": len(f), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                "hbm_bytes_per_launch": round(fb + wb), "launches_per_step": len(f) // STEPS}
# which library was profiled: bench.py quotes these bytes only for the same library SOURCES (animate_anything_amd.build.source_id)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd import build  # noqa: E402
out["library_source_sha256_16"] = build.source_id()
print(json.dumps(out, indent=1))
