#!/bin/bash
# r04 call R: the in-kernel K-split finish compiled into tiles 46 / 47 / 48 only (library 19 -> 13 MB): audit on the GPU side, tile fuzz
# with and without tickets, the whole GPU suite, one bench line
OUT=gpurun_out/r04r
mkdir -p $OUT
timeout 600 python scripts/debug/fuzz_tiles_fullsize.py 60 4 21 > $OUT/fuzz.log 2>&1; echo "tile fuzz rc=$? $(grep -c ' ok' $OUT/fuzz.log) ok $(grep -c FAIL $OUT/fuzz.log) fail" >> $OUT/summary.log
AA_TICKETS=1 timeout 600 python scripts/debug/fuzz_tiles_fullsize.py 60 4 22 > $OUT/fuzz_tickets.log 2>&1; echo "tile fuzz (tickets) rc=$? $(grep -c ' ok' $OUT/fuzz_tickets.log) ok $(grep -c FAIL $OUT/fuzz_tickets.log) fail" >> $OUT/summary.log
timeout 1700 python -m pytest tests -m gpu -q -n 3 > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $OUT/summary.log
tail -2 $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.log
timeout 900 python bench.py > $OUT/bench.json 2>$OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
head -c 300 $OUT/bench.json; echo
cat $OUT/summary.log
