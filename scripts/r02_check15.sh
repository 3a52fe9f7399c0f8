#!/bin/bash
# CLI end to end with the decoded frames resident in HBM; the per-frame GPU conversion with more hardware queues
mkdir -p gpurun_out/r02p; export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_gpu_io.py -m gpu -q --tb=short -x 2>&1 | tail -25) > gpurun_out/r02p/pytest.log 2>&1
tail -3 gpurun_out/r02p/pytest.log
run() { name=$1; shift; timeout 300 "$@" > gpurun_out/r02p/$name.log 2>&1; tail -1 gpurun_out/r02p/$name.log | cut -c1-420; }
run res200 python scripts/bench_cli.py --frames 200
run res600 python scripts/bench_cli.py --frames 600
run res600_444 python scripts/bench_cli.py --frames 600 --chroma-out 444
run frame300_q8 env GPU_MAX_HW_QUEUES=8 python scripts/bench_cli.py --frames 300 --resident 0
