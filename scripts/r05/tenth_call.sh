#!/bin/bash
# Round 5: RAFT run lanes (VSR_RAFT_LANES, default 2): same frames, config 4
OUT=gpurun_out/r05_tenth; mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_pp.py -q -k "raft_run_lanes or window_lanes" 2>&1 | tail -4) > $OUT/pytest.log; cat $OUT/pytest.log
for cfg in "2 4" "1 4" "2 4h" "1 4h"; do
  set -- $cfg
  (VSR_RAFT_LANES=$1 timeout 600 python scripts/bench_configs.py $2 2>&1 | grep '^{') > $OUT/raft_lanes$1_$2.json
  python -c "
import json; d=json.load(open('$OUT/raft_lanes$1_$2.json')); print('VSR_RAFT_LANES=$1 leg $2:', d.get('value'), 'fps', d.get('s_per_batch'), 's/batch', d.get('error'))"
done
(timeout 600 python -m pytest tests/test_gpu_zbaseline.py -q -s -k "config4 and (default or f16)" 2>&1 | grep -E "PSNR|passed|failed" | tail -14) > $OUT/pytest_config4.log; cat $OUT/pytest_config4.log
