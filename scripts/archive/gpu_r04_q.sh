#!/bin/bash
# r04 call Q: norm3 -> GEGLU fold on / off on one box (the fold costs the GEGLU projection more than the LayerNorm kernel it replaces?)
OUT=$PWD/gpurun_out/r04q
mkdir -p $OUT
AA_LN_FOLD_FF=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-form --tile-cache $OUT/tc_on.json --gemm-breakdown $OUT/gemm_on.txt > $OUT/bench_on.json 2> $OUT/on.err; echo "on rc=$?" >> $OUT/summary.log
AA_LN_FOLD_FF=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-form --tile-cache $OUT/tc_off.json --gemm-breakdown $OUT/gemm_off.txt > $OUT/bench_off.json 2> $OUT/off.err; echo "off rc=$?" >> $OUT/summary.log
AA_LN_FOLD_FF=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-form --no-roofline --tile-cache $OUT/tc_on.json > $OUT/bench_on2.json 2>/dev/null
AA_LN_FOLD_FF=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-form --no-roofline --tile-cache $OUT/tc_off.json > $OUT/bench_off2.json 2>/dev/null
for f in on off on2 off2; do head -c 250 $OUT/bench_$f.json | cut -c60-250; echo; done
cat $OUT/summary.log
