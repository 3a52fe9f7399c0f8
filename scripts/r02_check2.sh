#!/bin/bash
# round-2 second GPU pass: LaMa on the GPU, sttn-det diagnosis at growing batch lengths
mkdir -p gpurun_out/r02b
timeout 900 python -m pytest tests/test_gpu_lama.py -m gpu -q -s -x --durations=5 > gpurun_out/r02b/pytest_lama.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02b/pytest_lama.log
timeout 300 python scripts/bench_lama.py > gpurun_out/r02b/bench_lama.log 2>&1
timeout 300 python scripts/bench_lama.py --precision split >> gpurun_out/r02b/bench_lama.log 2>&1
timeout 900 python scripts/det_diag.py 12 21 47 > gpurun_out/r02b/det_diag.log 2>&1
tail -15 gpurun_out/r02b/pytest_lama.log; cat gpurun_out/r02b/bench_lama.log | tail -3; cat gpurun_out/r02b/det_diag.log | tail -40
