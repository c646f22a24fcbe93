#!/bin/bash
# round 5, call i: the existing switches re-measured on the round-5 library (the FF merge changed what surrounds them): norm3 -> GEGLU fold
# on / off, raw LayerNorm statistics in the consumer on / off; two alternating pairs each, tile choices tuned per variant first
OUT=gpurun_out/r05i
mkdir -p $OUT
export TMPDIR=/tmp
TC=$OUT/tile_cache.json
cp animate_anything_amd/tile_cache_gfx950.json $TC
for v in "AA_X=0" "AA_LN_FOLD_FF=0" "AA_LN_RAW=1"; do
env $v timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/tune.log 2>&1
done
for rep in 1 2; do
for v in "AA_X=0" "AA_LN_FOLD_FF=0" "AA_LN_RAW=1"; do
env $v timeout 600 python bench.py --no-cpu-baseline --no-vae --no-other-form --no-roofline --tile-cache $TC > $OUT/bench_${v}_$rep.json 2>$OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench_${v}_$rep.json')); print('$v', $rep, d['ms_per_step'], d['autotuned_signatures'])"
done
done
