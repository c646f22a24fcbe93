#!/bin/bash
# r04: per-kernel SQ counters of the final library - 3x3 convolution 320 -> 320 at 64x64 (autotuned tile) and the 4096-key attention
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash scripts/pmc_conv_sq.sh r04pmc_conv > /dev/null 2>&1
bash scripts/pmc_attn.sh r04pmc_attn > /dev/null 2>&1
cat gpurun_out/r04pmc_conv/summary.log gpurun_out/r04pmc_attn/summary.log
grep -B1 -A22 "conv_gemm_x_kernel" gpurun_out/r04pmc_conv/report.txt | head -60
grep -A22 "attention_kernel" gpurun_out/r04pmc_attn/report.txt | head -40
find gpurun_out/r04pmc_conv gpurun_out/r04pmc_attn -name "*.csv" -size +1M -delete
