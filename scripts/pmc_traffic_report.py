#!/usr/bin/env python
"""Reduce the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_traffic.sh to per-kernel-family HBM bytes per launch.
Units: rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE counts 128-byte requests of wide (16 B/lane)
coalesced reads at 64 B, i.e. HALF the bytes (MI355X_MICROARCH.md, HBM) -> fetch bytes are doubled here (upper
bound for narrow accesses).  Two steps are profiled (1 warm-up + 1 timed): per-launch means are step-independent."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]


def family(name):
    if "conv_gemm_dma_kernel" in name or "conv3x3_slab_kernel" in name or "conv_gemm_x_kernel" in name:
        return "contraction_kernels"                       # the LDS-DMA implicit-GEMM family (compiled, halo-slab and hand-scheduled tiles)
    for key in ("conv_gemm_kernel", "splitk_reduce", "attention_kernel", "attention_shortkv", "groupnorm_apply",
                "groupnorm_stats", "groupnorm_fused", "layernorm_kernel", "ln_finalize", "cfg_dpm_step"):
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


fetch, write = collect("FETCH_SIZE"), collect("WRITE_SIZE")
out = {}
for fam in sorted(set(fetch) | set(write)):
    f, w = fetch.get(fam, []), write.get(fam, [])
    fb = 2.0 * 1024.0 * sum(f) / max(len(f), 1)
    wb = 1024.0 * sum(w) / max(len(w), 1)
    out[fam] = {"launches_profiled": len(f), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                "hbm_bytes_per_launch": round(fb + wb), "launches_per_step": len(f) // 2}
# which library was profiled: bench.py quotes these bytes only for the same library SOURCES (animate_anything_amd.build.source_id)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate_anything_amd import build  # noqa: E402
out["library_source_sha256_16"] = build.source_id()
print(json.dumps(out, indent=1))
