#!/bin/bash
# BASELINE config 3 (sttn-det, detector on) file to file: the detector pass is 40 % of the run at 44 TF -- more detector lanes than the
# default 2 (round 4 measured 1 -> 2 only: 4.55 -> 4.05 s per 600 sampled frames), and the column ranges on the inpainting side.
OUT=gpurun_out/r05_detlanes; mkdir -p $OUT; CLIP=/tmp/vsr_e2e_clip_1080p_1200.y4m
for v in "2 0 8" "3 0 8" "4 0 8" "2 1 8"; do
  set -- $v
  (VSR_DET_LANES=$1 VSR_DECODE_COLS=$2 VSR_DET_BATCH=$3 timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 1200 --mode sttn-det --resident 1 2>&1 | tail -4) > $OUT/det_lanes$1_cols$2_batch$3.log
  grep '"metric"' $OUT/det_lanes$1_cols$2_batch$3.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('VSR_DET_LANES=$1 VSR_DECODE_COLS=$2 VSR_DET_BATCH=$3:', d['value'], 'fps', d['wall_s'], 's', d['phases_s'])"
done
