#!/bin/bash
# last call of round 6: the whole GPU suite, smoke() and the driver's bench command on the final tree
OUT=gpurun_out/r06_last; mkdir -p $OUT
(timeout 2400 python -m pytest tests -m gpu -q -x --tb=short --durations=5 2>&1 | tail -30) > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
(timeout 400 python -c "import __graft_entry__ as g; g.smoke()") > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
/usr/bin/time -v python bench.py > $OUT/bench.log 2> $OUT/bench.err; grep '"metric"' $OUT/bench.log | cut -c1-400; grep "Elapsed (wall" $OUT/bench.err
