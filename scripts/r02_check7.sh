#!/bin/bash
mkdir -p gpurun_out/r02g; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ocr_det.py -m gpu -q -x > gpurun_out/r02g/pytest_det.log 2>&1
tail -3 gpurun_out/r02g/pytest_det.log
timeout 900 python scripts/bench_det.py > gpurun_out/r02g/bench_det.log 2>&1
grep -E "forward|frames per" gpurun_out/r02g/bench_det.log
