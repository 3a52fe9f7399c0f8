#!/bin/bash
OUT=gpurun_out/r03_b1; mkdir -p $OUT
timeout 300 video-subtitle-remover_amd/build/v3_probe > $OUT/v3_probe.log 2>&1; grep -v "^      xcd [1-7]" $OUT/v3_probe.log | head -20
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sttn.py -m gpu -x -q 2>&1 | tail -5
python bench.py --no-cpu-baseline --e2e-chunks 0 --no-split-half > $OUT/bench.log 2>&1; grep '"metric"' $OUT/bench.log | cut -c1-300
