#!/bin/bash
# detector lanes default 2 + streaming store: the new GPU tests, then configs 3 and 4 file to file (scripts/bench_e2e.py)
OUT=gpurun_out/r04_e2e2; mkdir -p $OUT; CLIP=/tmp/vsr_e2e_clip_1080p_1200.y4m
(timeout 900 python -m pytest tests/test_gpu_ocr_det.py tests/test_gpu_io.py -q -x 2>&1 | tail -4) > $OUT/pytest.log; tail -2 $OUT/pytest.log
i=0
for run in "sttn-det --resident 1" "sttn-det --resident 1" "propainter --resident 1"; do
  i=$((i+1)); tag=$(echo $run | tr ' -' '__')_$i
  (timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 1200 --mode $run 2>&1 | tail -4) > $OUT/$tag.log; grep '"metric"' $OUT/$tag.log | cut -c1-600
done
