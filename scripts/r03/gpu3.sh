#!/bin/bash
OUT=gpurun_out/r03_probe; mkdir -p $OUT
timeout 300 video-subtitle-remover_amd/build/v3_probe > $OUT/v3_probe_c.log 2>&1; grep -v "^      xcd [1-7]" $OUT/v3_probe_c.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -5
