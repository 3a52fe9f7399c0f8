#!/bin/bash
# the N > 1 bench path (scatter -> inpaint on rank k -> gather, with the self-test of the exchanged bytes) on a 1-GPU box:
# two ranks share cuda:0, exchanges over gloo through host memory (VSR_BENCH_DRYRUN_1GPU=1); and the same with a corrupted
# exchange (VSR_BENCH_SELFTEST_CORRUPT=1) to see the self-test fail
OUT=gpurun_out/r03; mkdir -p $OUT
for corrupt in 0 1; do
  VSR_BENCH_DRYRUN_1GPU=1 VSR_BENCH_SELFTEST_CORRUPT=$corrupt timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 2951$corrupt bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/dryrun_2ranks_corrupt$corrupt.log 2>&1
  echo "corrupt=$corrupt rc=$?"; grep -o '"selftest": {[^}]*}' $OUT/dryrun_2ranks_corrupt$corrupt.log | cut -c1-400; grep "SELFTEST" $OUT/dryrun_2ranks_corrupt$corrupt.log
  grep -o '"value": [0-9.]*, "unit": "frames/s", "n_gpus": 2' $OUT/dryrun_2ranks_corrupt$corrupt.log
done
