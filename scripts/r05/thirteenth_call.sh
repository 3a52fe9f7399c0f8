#!/bin/bash
# config 4 file to file: do the generator / RAFT lanes help there (streams share hardware queues with the store / detector streams)?
OUT=gpurun_out/r05_thirteenth; mkdir -p $OUT; CLIP=/tmp/vsr_e2e_clip_1080p_600_on.y4m
run() {   # run NAME ENV...
  local name=$1; shift
  (env "$@" timeout 600 python scripts/bench_e2e.py --clip $CLIP --frames 600 --always-on --mode propainter 2>&1 | grep '"metric"') > $OUT/e2e_$name.json
  python -c "
import json; d=json.load(open('$OUT/e2e_$name.json')); print('$name:', d['value'], 'fps', d['wall_s'], 's', d['phases_s']['inpainting'], 's inpainting')"
}
run lanes1 VSR_PP_LANES=1 VSR_RAFT_LANES=1
run lanes2 VSR_PP_LANES=2 VSR_RAFT_LANES=2
run lanes2_hwq8 VSR_PP_LANES=2 VSR_RAFT_LANES=2 GPU_MAX_HW_QUEUES=8
