#!/bin/bash
# config 3 file to file: frames per detector forward 8 (default) vs 16, with the default two detector lanes
OUT=gpurun_out/r04_detb; mkdir -p $OUT; CLIP=/tmp/vsr_e2e_clip_1080p_1200.y4m
for b in 8 16 8 16; do
  (VSR_DET_BATCH=$b timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 1200 --mode sttn-det --resident 1 2>&1 | tail -4) > $OUT/b$b.log
  grep '"metric"' $OUT/b$b.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('VSR_DET_BATCH=$b:', d['value'], 'fps', d['wall_s'], 's', d['phases_s'], d['detector']['forwards'], 'forwards')"
done
