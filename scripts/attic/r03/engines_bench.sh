#!/bin/bash
# round 3: the other engines on the rebuilt gather-GEMM (detector, flow completion, LaMa, RAFT, ProPainter plugin) + the default bench line
OUT=gpurun_out/r03_engines; mkdir -p $OUT; export TMPDIR=/tmp
for b in det rfc lama raft propainter; do
  (timeout 600 python scripts/bench_$b.py 2>&1 | tail -25) > $OUT/bench_$b.log 2>&1; echo "== $b"; tail -6 $OUT/bench_$b.log | cut -c1-300
done
(timeout 900 python bench.py 2>&1) > $OUT/bench.log; grep '"metric"' $OUT/bench.log | cut -c1-250
