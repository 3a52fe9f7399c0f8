#!/bin/bash
# round 6, seventeenth call: RAFT workspace cleared by extent -- the suite (with the alternation test), config 4 / 4h lines
OUT=gpurun_out/r06_seventeenth; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_raft.py tests/test_gpu_pp.py -q -x 2>&1 | tail -4) > $OUT/pytest.log; cat $OUT/pytest.log
python scripts/bench_configs.py 4 4h > $OUT/configs_4.log 2>&1; grep '^{' $OUT/configs_4.log | cut -c1-330
