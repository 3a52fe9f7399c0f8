#!/bin/bash
# Round 5: the ProPainter generator's decoder box (vsr_pp_forward_box) was built and replayed on the CPU in round 4 but
# never ran on a GPU; neither did the per-frame encoder cache (vsr_pp_encode / vsr_pp_forward_cached, VSR_PP_ENC_CACHE=1: -25 % generator
# FLOPs).  1. bit equality at the 1080p strip size; 2. the plugin against the oracle with the switches on;
# 3. BASELINE config 4 file to file with the switches off / on (600 frames each keep the call short).
# If 1-2 are green and 3 follows the FLOPs (9 % + 25 % of the generator): make both switches default 1 (vsr_amd/switches.py).
OUT=gpurun_out/r05_ppbox; mkdir -p $OUT; CLIP=/tmp/vsr_e2e_clip_1080p_600.y4m
(VSR_PP_DECODE_BOX=1 VSR_PP_ENC_CACHE=1 timeout 900 python -m pytest tests/test_gpu_pp.py -q -x -k "decoder_box or encoder_cache or plugin_matches" 2>&1 | tail -5) > $OUT/pytest.log; tail -2 $OUT/pytest.log
(VSR_PP_DECODE_BOX=1 VSR_PP_ENC_CACHE=1 timeout 900 python -m pytest tests/test_gpu_golden_wrappers.py -q -x -k propainter 2>&1 | tail -3) >> $OUT/pytest.log; tail -1 $OUT/pytest.log
for v in "0 0" "1 1" "1 0" "0 1"; do
  set -- $v
  (VSR_PP_DECODE_BOX=$1 VSR_PP_ENC_CACHE=$2 timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 600 --mode propainter --resident 1 2>&1 | tail -4) > $OUT/pp_box$1_cache$2.log
  grep '"metric"' $OUT/pp_box$1_cache$2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('VSR_PP_DECODE_BOX=$1 VSR_PP_ENC_CACHE=$2:', d['value'], 'fps', d['wall_s'], 's', d['phases_s'])"
done
