#!/bin/bash
# round 6, fourth call: the weight-statistics sweep on the GPU (every failing case reported, none stops the run)
OUT=gpurun_out/r06_fourth; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_weight_sweep.py -q -m gpu -s 2>&1 | grep -E "(sttn|generator|flow completion|raft) \[|passed|failed|FAILED" | sed 's/^[.F]*//' > $OUT/pytest_sweep.log; cat $OUT/pytest_sweep.log
