#!/bin/bash
# round 6, call h: fused temporal attention with the pass bias added behind the pass (no load in front of the first MFMA), fragment reads two k-slices
# ahead, operand-slot reads batched; the two-query-block attention kernel behind its flag
OUT=gpurun_out/r06h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_seq_attention.py tests/test_kernels.py -m gpu -q -x --tb=short -k "seq or attention" > $OUT/test_seq.log 2>&1; echo "seq + attention tests rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_seq_attention.py > $OUT/bench_seq.log 2>&1; echo "bench_seq rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_seq_attention.py --ablate > $OUT/ablate.log 2>&1; echo "ablate rc=$?" >> $OUT/summary.log
TC=$OUT/tile_cache.json
timeout 1500 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
AA_SEQ_ATTN=0 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune3.log 2>&1
for rep in 1 2 3; do
AA_SEQ_ATTN=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_off_$rep.json 2>$OUT/bench.err; echo "bench off rc=$?" >> $OUT/summary.log
AA_SEQ_ATTN=1 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_on_$rep.json 2>$OUT/bench.err; echo "bench on rc=$?" >> $OUT/summary.log
done
cat $OUT/summary.log
tail -3 $OUT/test_seq.log
grep "C=" $OUT/bench_seq.log
grep "C=" $OUT/ablate.log
for f in $OUT/bench_off_1.json $OUT/bench_on_1.json $OUT/bench_off_2.json $OUT/bench_on_2.json $OUT/bench_off_3.json $OUT/bench_on_3.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
