#!/bin/bash
OUT=gpurun_out/r03b
mkdir -p $OUT
timeout 900 python scripts/x_ablate.py > $OUT/x_ablate.log 2>&1; echo "ablate rc=$?" >> $OUT/summary.log
cat $OUT/summary.log; cat $OUT/x_ablate.log
