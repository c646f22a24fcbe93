#!/bin/bash
OUT=gpurun_out/r03t
mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $OUT/summary.log
tail -15 $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.log
tail -3 $OUT/smoke.log; cat $OUT/summary.log
