#!/bin/bash
# Round 4, first GPU call: the detector lanes / batch lanes built unmeasured at the end of round 3, on config 3 (and config 4's batch lanes),
# file to file through SubtitleRemover.run() (scripts/bench_e2e.py); then the default bench line of the unchanged tree as this round's baseline.
OUT=gpurun_out/r04_lanes; mkdir -p $OUT; CLIP=/tmp/vsr_e2e_clip_1080p_1200.y4m
run() { # tag, env..., -- args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  (env "${envs[@]}" timeout 600 python scripts/bench_e2e.py --clip $CLIP --frames 1200 "$@" 2>&1 | tail -4) > $OUT/$tag.log
  grep '"metric"' $OUT/$tag.log | cut -c1-700
}
run det_l1_b1 VSR_DET_LANES=1 VSR_BATCH_LANES=1 -- --mode sttn-det
run det_l2_b1 VSR_DET_LANES=2 VSR_BATCH_LANES=1 -- --mode sttn-det
run det_l1_b2 VSR_DET_LANES=1 VSR_BATCH_LANES=2 -- --mode sttn-det
run det_l2_b2 VSR_DET_LANES=2 VSR_BATCH_LANES=2 -- --mode sttn-det
run det_l3_b2 VSR_DET_LANES=3 VSR_BATCH_LANES=2 -- --mode sttn-det
run pp_l2_b2 VSR_DET_LANES=2 VSR_BATCH_LANES=2 -- --mode propainter
python bench.py --no-cpu-baseline > $OUT/bench.log 2>&1; grep '"metric"' $OUT/bench.log | cut -c1-300
