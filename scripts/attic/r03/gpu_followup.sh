#!/bin/bash
OUT=gpurun_out/r03; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ocr_det.py -m gpu -q -k "batched or recorded or plumbing" 2>&1 | tail -3
bash scripts/r03/dryrun_2ranks.sh
python bench.py > $OUT/bench2.log 2>&1; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r03/bench2.log') if x.startswith('{"metric"')][0]
d=json.loads(l); print(d['value'], d['roofline']['frac'], json.dumps(d['cpu_baseline'].get('parallel'))[:900])
PY
