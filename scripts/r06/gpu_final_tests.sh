#!/bin/bash
# round 6, end of round: the whole GPU suite + smoke() with the final library
OUT=gpurun_out/r06_tests; mkdir -p $OUT
(timeout 2400 python -m pytest tests -m gpu -q -x --tb=short --durations=8 2>&1 | tail -60) > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
(timeout 400 python -c "import __graft_entry__ as g; g.smoke()") > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
