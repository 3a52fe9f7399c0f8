#!/bin/bash
# BASELINE configs 3 and 4 as stated, through SubtitleRemover.run(), file to file (scripts/bench_e2e.py)
OUT=gpurun_out/r03_e2e; mkdir -p $OUT; CLIP=/tmp/vsr_e2e_clip_1080p_1200.y4m
for run in "sttn-det --resident 1" "sttn-det --resident 1" "sttn-det --resident 0" "propainter --resident 1" "propainter --resident 1 --precision split"; do
  tag=$(echo $run | tr ' -' '__')
  (timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 1200 --mode $run 2>&1 | tail -4) > $OUT/$tag.log; grep '"metric"' $OUT/$tag.log | cut -c1-900
done
