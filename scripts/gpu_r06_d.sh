#!/bin/bash
# round 6, call d: fused temporal attention with the output stores left in flight across the head boundary and the two workgroups of a CU
# started out of phase: parity, start-offset sweep, ablations again, step A/B
OUT=gpurun_out/r06d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_seq_attention.py -m gpu -q -x --tb=short > $OUT/test_seq.log 2>&1; echo "seq tests rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_seq_attention.py --dephase > $OUT/dephase.log 2>&1; echo "dephase rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_seq_attention.py --ablate > $OUT/ablate.log 2>&1; echo "ablate rc=$?" >> $OUT/summary.log
timeout 600 python scripts/bench_seq_attention.py > $OUT/bench_seq.log 2>&1; echo "bench_seq rc=$?" >> $OUT/summary.log
for rep in 1 2; do
AA_SEQ_ATTN=0 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline > $OUT/bench_off_$rep.json 2>$OUT/bench.err; echo "bench off rc=$?" >> $OUT/summary.log
AA_SEQ_ATTN=1 timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline > $OUT/bench_on_$rep.json 2>$OUT/bench.err; echo "bench on rc=$?" >> $OUT/summary.log
done
cat $OUT/summary.log
tail -3 $OUT/test_seq.log
grep -v amdgpu.ids $OUT/dephase.log | grep "dephase\|C=320 rows=139264"
cat $OUT/ablate.log | grep "C=320"
cat $OUT/bench_seq.log
for f in $OUT/bench_off_1.json $OUT/bench_on_1.json $OUT/bench_off_2.json $OUT/bench_on_2.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['autotuned_signatures'])"; done
