#!/bin/bash
# round 6, eighth call: the whole GPU suite and smoke() on the tree after the v8 removal / ADVICE fixes
OUT=gpurun_out/r06_eighth; mkdir -p $OUT; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > $OUT/smoke.log; cat $OUT/smoke.log
