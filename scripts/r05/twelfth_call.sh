#!/bin/bash
# config 4 file to file (600 frames, subtitle on every frame -> 70-frame batches) with the lanes of the final tree, lane instances built
# on a helper thread; the plugin tests again
OUT=gpurun_out/r05_twelfth; mkdir -p $OUT; CLIP=/tmp/vsr_e2e_clip_1080p_600_on.y4m
(timeout 300 python -m pytest tests/test_gpu_pp.py tests/test_gpu_golden_wrappers.py -q -k "lanes or plugin_matches or propainter" 2>&1 | tail -2)
for p in f32 f16; do
  (timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 600 --always-on --mode propainter --precision $p 2>&1 | grep '"metric"') > $OUT/e2e_pp_$p.json
  python -c "
import json; d=json.load(open('$OUT/e2e_pp_$p.json')); print('config 4 file to file, 600 frames always on, $p:', d['value'], 'fps', d['wall_s'], 's', d['phases_s'])"
done
