#!/bin/bash
# Round 5: the ProPainter generator's decoder box (vsr_pp_forward_box) was built and replayed on the CPU in round 4 but
# never ran on a GPU.  1. bit equality inside the box at the 1080p strip size; 2. the plugin against the oracle with the switch on;
# 3. BASELINE config 4 file to file with the switch off / on (600 frames each keep the call short).
# If 1-2 are green and 3 follows the FLOPs (9 % of the generator): make VSR_PP_DECODE_BOX default 1 (propainter_inpaint.py).
OUT=gpurun_out/r05_ppbox; mkdir -p $OUT; CLIP=/tmp/vsr_e2e_clip_1080p_600.y4m
(VSR_PP_DECODE_BOX=1 timeout 900 python -m pytest tests/test_gpu_pp.py -q -x -k "decoder_box or plugin_matches" 2>&1 | tail -5) > $OUT/pytest.log; tail -2 $OUT/pytest.log
(VSR_PP_DECODE_BOX=1 timeout 900 python -m pytest tests/test_gpu_golden_wrappers.py -q -x -k propainter 2>&1 | tail -3) >> $OUT/pytest.log; tail -1 $OUT/pytest.log
for v in 0 1 0 1; do
  (VSR_PP_DECODE_BOX=$v timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 600 --mode propainter --resident 1 2>&1 | tail -4) > $OUT/pp_box$v.log
  grep '"metric"' $OUT/pp_box$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('VSR_PP_DECODE_BOX=$v:', d['value'], 'fps', d['wall_s'], 's', d['phases_s'])"
done
