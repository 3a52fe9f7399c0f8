#!/bin/bash
# sttn-det with the decoder rows (vsr_sttn_det_batch_rows): STTN engine + golden wrapper + IO tests, configs resident, config 3 file to file
OUT=gpurun_out/r04_rowsdet; mkdir -p $OUT; CLIP=/tmp/vsr_e2e_clip_1080p_1200.y4m
(timeout 900 python -m pytest tests/test_gpu_sttn.py tests/test_gpu_golden_wrappers.py tests/test_gpu_io.py -q -x 2>&1 | tail -3) > $OUT/pytest.log; tail -1 $OUT/pytest.log
(timeout 400 python scripts/bench_configs.py 2>&1 | grep '^{') > $OUT/configs.log; cut -c1-230 $OUT/configs.log
for v in 1 0 1; do
  (VSR_DECODE_ROWS=$v timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 1200 --mode sttn-det --resident 1 2>&1 | tail -4) > $OUT/e2e_rows$v.log
  grep '"metric"' $OUT/e2e_rows$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('VSR_DECODE_ROWS=$v:', d['value'], 'fps', d['wall_s'], 's', d['phases_s'])"
done
