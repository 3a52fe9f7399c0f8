#!/bin/bash
# Round 5: the server detector's 64 -> 64 2x2 transposed conv as a gather-GEMM (was a direct kernel: 5.3 ms of a 16-frame forward)
OUT=gpurun_out/r05_seventh; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_ocr_det.py -q -s -k "not db_postprocess and not hole_count" 2>&1 | grep -E "err vs|passed|failed|Error|error" | tail -14) > $OUT/pytest_det.log; cat $OUT/pytest_det.log
for v in 1 0 1 0; do
  (VSR_DET_DECONV_GEMM=$v timeout 300 python scripts/bench_configs.py 3d 2>&1 | grep '^{') > $OUT/det_deconv_gemm$v.json
  python -c "
import json; d=json.load(open('$OUT/det_deconv_gemm$v.json')); print('VSR_DET_DECONV_GEMM=$v:', d.get('ms_per_frame'), 'ms/frame', d.get('value'), 'fps', (d.get('roofline') or {}).get('frac'), d.get('error'))"
done
