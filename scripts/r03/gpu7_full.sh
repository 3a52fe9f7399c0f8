#!/bin/bash
# the whole GPU suite after the kernel / post-process / plugin changes of round 3
OUT=gpurun_out/r03_full; mkdir -p $OUT
(timeout 2400 python -m pytest tests -m gpu -q --tb=short --durations=10 2>&1 | tail -40) > $OUT/pytest_gpu.log 2>&1
tail -25 $OUT/pytest_gpu.log
