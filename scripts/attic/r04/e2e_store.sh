#!/bin/bash
# config 3 (and 4) file to file with the streaming store (frames written under the batches that follow): GPU IO tests, then
# scripts/bench_e2e.py; VSR_DET_LANES=2 beside the default
OUT=gpurun_out/r04_e2e; mkdir -p $OUT; CLIP=/tmp/vsr_e2e_clip_1080p_1200.y4m
(timeout 600 python -m pytest tests/test_gpu_io.py -q -x 2>&1 | tail -3) > $OUT/pytest_io.log; tail -1 $OUT/pytest_io.log
i=0
for run in "sttn-det --resident 1" "sttn-det --resident 1" "DL2 sttn-det --resident 1" ${E2E_MORE:+"propainter --resident 1"}; do
  i=$((i+1)); tag=$(echo $run | tr ' -' '__')_$i
  if [ "${run%% *}" = "DL2" ]; then run=${run#DL2 }; export VSR_DET_LANES=2; else unset VSR_DET_LANES; fi
  (timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 1200 --mode $run 2>&1 | tail -4) > $OUT/$tag.log; grep '"metric"' $OUT/$tag.log | cut -c1-700
done
