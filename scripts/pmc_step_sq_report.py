#!/usr/bin/env python
"""Reduce the SQ passes of scripts/pmc_step_sq.sh to per-kernel-family sums over one eager step (two steps are profiled: halved).
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the SQ counter adds up the cycles each SIMD's matrix
pipe was busy, and rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (476 contraction launches: 854.9 M = 8 x the 106.9 M
shader cycles of their 49.8 ms); valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA (wave instructions)."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]


def family(name):
    if "conv_gemm_dma_kernel" in name or "conv3x3_slab_kernel" in name or "conv_gemm_x_kernel" in name:
        return "contraction_kernels"
    for key in ("ff_fused_kernel", "linear_rows_kernel", "seq_self_attention_kernel", "conv_gemm_kernel", "splitk_reduce", "attention_kernel", "attention_shortkv",
                "groupnorm_apply", "groupnorm_stats", "groupnorm_fused", "groupnorm_coef", "layernorm_kernel", "ln_finalize"):
        if key in name:
            return key
    return None


acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for f in glob.glob(f"{root}/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        fam = family(r["Kernel_Name"])
        if fam:
            acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_WAVES"):
                n[(fam, r["Counter_Name"])] += 1
out = {}
for fam, c in sorted(acc.items()):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    rec = {k: round(v / 2) for k, v in sorted(c.items())}
    rec["launches_per_step"] = n[(fam, "GRBM_GUI_ACTIVE")] // 2
    if gui and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        rec["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0), 4)
    if c.get("SQ_INSTS_MFMA"):
        rec["valu_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"], 2)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        rec["lds_bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    out[fam] = rec
print(json.dumps(out, indent=1))
