#!/bin/bash
mkdir -p gpurun_out/r02r; export TMPDIR=/tmp
(timeout 400 python -m pytest tests/test_gpu_io.py -m gpu -q --tb=short -x --durations=4 2>&1 | tail -25) > gpurun_out/r02r/pytest.log 2>&1
tail -8 gpurun_out/r02r/pytest.log
