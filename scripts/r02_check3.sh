#!/bin/bash
# round-2 third GPU pass: propainter plugin with the on-device glue, det / propainter BASELINE-size parity, LaMa kernel trace
mkdir -p gpurun_out/r02c; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pp.py tests/test_gpu_multirank.py "tests/test_gpu_zbaseline.py::test_det_batch_L47_vs_oracle" "tests/test_gpu_zbaseline.py::test_propainter_batch_L20_vs_oracle" -m gpu -q -s --durations=8 > gpurun_out/r02c/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02c/pytest.log
timeout 300 python scripts/bench_propainter.py > gpurun_out/r02c/bench_propainter.log 2>&1
timeout 300 python scripts/bench_propainter.py --precision split >> gpurun_out/r02c/bench_propainter.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02c/lama_trace -o r -- python scripts/bench_lama.py --steps 3 --warmup 1 > gpurun_out/r02c/lama_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02c/pp_trace -o r -- python scripts/bench_propainter.py > gpurun_out/r02c/pp_trace.log 2>&1
rm -f gpurun_out/r02c/*/r_kernel_trace.csv gpurun_out/r02c/*/*/r_kernel_trace.csv
grep -E "PSNR|passed|failed|FAILED|Error|error" gpurun_out/r02c/pytest.log | tail -20; cat gpurun_out/r02c/bench_propainter.log | grep metric; find gpurun_out/r02c -name "*kernel_stats.csv" | head
