#!/bin/bash
# two window lanes (vsr_sttn_set_lanes): bit-equality tests, then the bench with 1 and 2 lanes
OUT=gpurun_out/r03_lanes; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sttn.py -m gpu -x -q -k lanes 2>&1 | tail -4
for l in 2 3 4; do
  python bench.py --lanes $l --no-cpu-baseline --e2e-chunks 0 > $OUT/bench_lanes$l.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/r03_lanes/bench_lanes$l.log') if x.startswith('{"metric"')]
if not l: print(open('gpurun_out/r03_lanes/bench_lanes$l.log').read()[-2000:])
else:
    d=json.loads(l[0]); print('lanes $l:', d['value'], 'fps; single-lane leg', d['single_lane']['value'], '; roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'],
      '; split', d['split_half_mode']['value'], d['split_format_mode']['value'], 'f16', d['fp16_mode']['value'])
PY
done
