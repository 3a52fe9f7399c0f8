#!/bin/bash
# Round 5: generator window lanes, default 2 (threaded in the guarded arithmetics): same frames, config 4 in both arithmetics, parity
OUT=gpurun_out/r05_ninth; mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_pp.py tests/test_gpu_golden_wrappers.py -q -k "window_lanes or plugin_matches or propainter" 2>&1 | tail -4) > $OUT/pytest_lanes2.log; cat $OUT/pytest_lanes2.log
for cfg in "2 4" "1 4" "2 4h" "1 4h" "2 4s"; do
  set -- $cfg
  (VSR_PP_LANES=$1 timeout 600 python scripts/bench_configs.py $2 2>&1 | grep '^{') > $OUT/pp_lanes$1_$2.json
  python -c "
import json; d=json.load(open('$OUT/pp_lanes$1_$2.json')); print('VSR_PP_LANES=$1 leg $2:', d.get('value'), 'fps', d.get('s_per_batch'), 's/batch', d.get('psnr_db_vs_exact_mode'), d.get('error'))"
done
(timeout 600 python -m pytest tests/test_gpu_zbaseline.py -q -s -k "config4" 2>&1 | grep -E "PSNR|passed|failed" | tail -18) > $OUT/pytest_config4.log; cat $OUT/pytest_config4.log
