#!/bin/bash
# round 5, call g: lazy row maximum (the library) against the eager form of rounds 2-4 (libaa_mi355_eagermax.so, -DAA_ATTN_EAGER_MAX=1:
# same source otherwise), both without register spills: per-kernel and per-step A/B on one box, alternating
OUT=gpurun_out/r05g
mkdir -p $OUT
export TMPDIR=/tmp
E=$PWD/animate_anything_amd/libaa_mi355_eagermax.so
for rep in 1 2; do
AA_LIBRARY=$E timeout 600 python scripts/bench_kernels.py --only "attn spatial" --reps 20 > $OUT/attn_eager_$rep.log 2>&1
timeout 600 python scripts/bench_kernels.py --only "attn spatial" --reps 20 > $OUT/attn_lazy_$rep.log 2>&1
done
for rep in 1 2 3; do
AA_LIBRARY=$E timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline > $OUT/bench_eager_$rep.json 2>$OUT/bench.err; echo "bench eager rc=$?" >> $OUT/summary.log
timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline > $OUT/bench_lazy_$rep.json 2>$OUT/bench.err; echo "bench lazy rc=$?" >> $OUT/summary.log
done
cat $OUT/summary.log
for f in $OUT/attn_*.log; do echo $f; grep -i "attn" $f; done
for f in $OUT/bench_eager_1.json $OUT/bench_lazy_1.json $OUT/bench_eager_2.json $OUT/bench_lazy_2.json $OUT/bench_eager_3.json $OUT/bench_lazy_3.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
