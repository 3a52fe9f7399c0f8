#!/bin/bash
# N <= 64 convs of the flow engines: 128x64 (VSR_N64_TILE=3, default) vs 256x64 (=2)
OUT=gpurun_out/r03_n64e; mkdir -p $OUT
for b in rfc raft; do for t in 3 2; do
  (VSR_N64_TILE=$t timeout 100 python scripts/bench_$b.py 2>&1 | grep '"metric"' | cut -c1-260) > $OUT/${b}_$t.log; echo "$b VSR_N64_TILE=$t: $(cat $OUT/${b}_$t.log)"
done; done
