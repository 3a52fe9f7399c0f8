#!/bin/bash
# Round 5, first GPU call: the column ranges of DESIGN 4.3c were built and replayed on the CPU in round 4 but never ran on a GPU.
#   1. bit equality of the frames with the promise-free call (tests/test_gpu_sttn.py::test_decoder_box_gives_the_same_frames)
#   2. the default bench with the columns on / off, interleaved on one box (fps, GFLOP per frame, dominant-kernel rate)
# If 1 is green and 2 follows the FLOPs: make VSR_DECODE_COLS default 1 (vsr_amd/switches.py).
OUT=gpurun_out/r05_cols; mkdir -p $OUT
(VSR_DECODE_COLS=1 timeout 900 python -m pytest tests/test_gpu_sttn.py -q -x -k "decoder_box or decoder_rows" 2>&1 | tail -5) > $OUT/pytest.log; tail -2 $OUT/pytest.log
# config 3's inpainting with the columns off / on: scripts/r04/rows_det.sh with VSR_DECODE_COLS=0 / 1 is the det-side A/B
B="python bench.py --no-cpu-baseline --no-configs --no-split-half --no-full-work --e2e-chunks 0 --steps 8 --warmup 2"
for i in 1 2; do
  for v in 1 0; do
    VSR_DECODE_COLS=$v timeout 600 $B > $OUT/bench_cols${v}_$i.log 2>&1
    grep '"metric"' $OUT/bench_cols${v}_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('COLS=$v run $i:', d['value'], 'fps', d['ms_per_step'], 'ms; single lane', d['single_lane']['value'], '; GFLOP/frame', d['gflop_per_frame'], '|', d.get('gflop_per_frame_reference'), '; model TF', d['model_tflops'], '; roofline', r['achieved'], r['frac'])"
  done
done
# the first block's q/k/v once per frame of the chunk instead of once per window (VSR_QKV0_SHARED, -0.6 % of a chunk's FLOPs): same bits?
(VSR_QKV0_SHARED=1 timeout 900 python -m pytest tests/test_gpu_sttn.py -q -x -k "shared_first" 2>&1 | tail -3) >> $OUT/pytest.log; tail -1 $OUT/pytest.log
for i in 1 2; do
  for v in 1 0; do
    VSR_QKV0_SHARED=$v timeout 600 $B > $OUT/bench_qkv0_${v}_$i.log 2>&1
    grep '"metric"' $OUT/bench_qkv0_${v}_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('QKV0_SHARED=$v run $i:', d['value'], 'fps', d['ms_per_step'], 'ms; GFLOP/frame', d['gflop_per_frame'])"
  done
done
# window lanes after the dead-work elimination made the last block's launches shorter: 2 (default) vs 3
for l in 2 3; do
  timeout 600 $B --lanes $l > $OUT/bench_lanes${l}.log 2>&1
  grep '"metric"' $OUT/bench_lanes${l}.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lanes $l:', d['value'], 'fps', d['ms_per_step'], 'ms')"
done
