#!/bin/bash
OUT=gpurun_out/r03j
mkdir -p $OUT
X_ABLS=0,32,64,96 X_ONLY="L2" X_CFGS=36,40,41,43 timeout 900 python scripts/x_ablate.py > $OUT/x_ablate_sync.log 2>&1; echo "rc=$?" >> $OUT/summary.log
cat $OUT/x_ablate_sync.log | grep -v amdgpu
