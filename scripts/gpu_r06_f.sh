#!/bin/bash
# round 6, call f: fused temporal attention with the LayerNorm's gamma / beta folded into weights / bias (the kernel only normalises rows, packed arithmetic)
# and the linear layer in front inside the kernel (proj_in / the previous layer's to_out + residual): AA_SEQ_PRE=0/1, AA_SEQ_ATTN=0/1
OUT=gpurun_out/r06f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_seq_attention.py -m gpu -q -x --tb=short > $OUT/test_seq.log 2>&1; echo "seq tests rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_seq_attention.py > $OUT/bench_seq.log 2>&1; echo "bench_seq rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_seq_attention.py --ablate > $OUT/ablate.log 2>&1; echo "ablate rc=$?" >> $OUT/summary.log
TC=$OUT/tile_cache.json
timeout 1500 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --tile-cache $TC > $OUT/tune.log 2>&1; echo "tune rc=$?" >> $OUT/summary.log
AA_SEQ_PRE=0 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune2.log 2>&1
AA_SEQ_ATTN=0 timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune3.log 2>&1
for rep in 1 2; do
AA_SEQ_ATTN=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_off_$rep.json 2>$OUT/bench.err; echo "bench off rc=$?" >> $OUT/summary.log
AA_SEQ_PRE=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_nopre_$rep.json 2>$OUT/bench.err; echo "bench nopre rc=$?" >> $OUT/summary.log
AA_SEQ_ATTN=1 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_on_$rep.json 2>$OUT/bench.err; echo "bench on rc=$?" >> $OUT/summary.log
done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_unet.py -x -q -s -k "metric_configuration or three_steps or small" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/summary.log
cat $OUT/summary.log
tail -3 $OUT/test_seq.log
grep "C=" $OUT/bench_seq.log
grep "C=" $OUT/ablate.log
for f in $OUT/bench_off_1.json $OUT/bench_nopre_1.json $OUT/bench_on_1.json $OUT/bench_off_2.json $OUT/bench_nopre_2.json $OUT/bench_on_2.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
tail -4 $OUT/tests.log
