#!/bin/bash
mkdir -p gpurun_out/r02f; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ocr_det.py -m gpu -q -x -s > gpurun_out/r02f/pytest_det.log 2>&1
tail -4 gpurun_out/r02f/pytest_det.log
timeout 600 python scripts/bench_det.py > gpurun_out/r02f/bench_det.log 2>&1
grep forward gpurun_out/r02f/bench_det.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02f/det_trace -o r -- python scripts/bench_det.py ppocr_det_graph.json > gpurun_out/r02f/det_trace.log 2>&1
rm -f gpurun_out/r02f/det_trace/r_kernel_trace.csv gpurun_out/r02f/det_trace/*/r_kernel_trace.csv
