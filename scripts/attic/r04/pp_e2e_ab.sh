#!/bin/bash
# config 4 file to file: where did 63.0 s (round 3) -> 67.5 s of inpainting come from?  detector lanes 1 / 2, streaming store on / off
OUT=gpurun_out/r04_ppab; mkdir -p $OUT; CLIP=/tmp/vsr_e2e_clip_1080p_1200.y4m
for v in "1 0" "2 1" "1 1"; do
  set -- $v
  (VSR_DET_LANES=$1 VSR_STREAM_STORE=$2 timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 1200 --mode propainter --resident 1 2>&1 | tail -4) > $OUT/pp_det$1_stream$2.log
  grep '"metric"' $OUT/pp_det$1_stream$2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('det lanes $1 stream store $2:', d['value'], 'fps', d['wall_s'], 's', d['phases_s'])"
done
