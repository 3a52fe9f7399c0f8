#!/bin/bash
# y4m transport: GPU tests (batched colour conversion, resident chunk loop), CLI end to end with host frames converted in batches
mkdir -p gpurun_out/r02q; export TMPDIR=/tmp
(timeout 400 python -m pytest tests/test_gpu_io.py -m gpu -q --tb=short -x 2>&1 | tail -25) > gpurun_out/r02q/pytest.log 2>&1
tail -3 gpurun_out/r02q/pytest.log
run() { name=$1; shift; timeout 300 "$@" > gpurun_out/r02q/$name.log 2>&1; tail -1 gpurun_out/r02q/$name.log | cut -c1-420; }
run frame300 python scripts/bench_cli.py --frames 300 --resident 0
run res600 python scripts/bench_cli.py --frames 600
