#!/bin/bash
# r04 call A: the 8-phase GEMM probe (all variants) + the library's own tiles on the same pure GEMMs, same box
OUT=$PWD/gpurun_out/r04a
mkdir -p $OUT
rocm-smi --showclocks --showpower > $OUT/smi_before.txt 2>&1
for v in base nostagger nosetprio st16x32 mfma32 bf16; do
  echo "=== $v" >> $OUT/probe.log
  timeout 300 scripts/probe/bin/gemm_8phase_$v >> $OUT/probe.log 2>&1; echo "rc=$?" >> $OUT/probe.log
done
# second pass of the base variant (run-to-run spread on this box)
echo "=== base (second run)" >> $OUT/probe.log
timeout 120 scripts/probe/bin/gemm_8phase_base 4096 4096 4096 8192 8192 8192 8704 1280 11520 >> $OUT/probe.log 2>&1
timeout 600 python scripts/gemm_library_ab.py > $OUT/library_ab.log 2>&1; echo "rc=$?" >> $OUT/library_ab.log
tail -n 60 $OUT/probe.log; tail -n 50 $OUT/library_ab.log
