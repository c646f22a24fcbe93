#!/bin/bash
OUT=gpurun_out/r03g
mkdir -p $OUT
X_ABLS=0,8,16,7,15,23,31 timeout 900 python scripts/x_ablate.py > $OUT/x_ablate_epilogue.log 2>&1; echo "rc=$?" >> $OUT/summary.log
grep -E "geglu|ff2 L0" $OUT/x_ablate_epilogue.log
