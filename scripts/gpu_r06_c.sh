#!/bin/bash
# round 6, call c: where the fused temporal attention kernel's time goes (AaSeqSelfAttn.flags ablations); does v_exp_f32 block the VALU?
OUT=gpurun_out/r06c; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/bench_seq_attention.py --ablate > $OUT/ablate.log 2>&1; echo "ablate rc=$?" >> $OUT/summary.log
timeout 120 scripts/probe/bin/trans_rate > $OUT/trans_rate.log 2>&1; echo "trans_rate rc=$?" >> $OUT/summary.log
cat $OUT/summary.log; cat $OUT/ablate.log; cat $OUT/trans_rate.log
