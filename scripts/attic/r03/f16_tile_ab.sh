#!/bin/bash
OUT=gpurun_out/r03_f16; mkdir -p $OUT
VSR_CONV_TILE=4 timeout 900 python -m pytest tests/test_gpu_sttn.py -m gpu -x -q -k "f16 or split or precision or range" 2>&1 | tail -4
for t in 3 4; do
VSR_CONV_TILE=$t python bench.py --precision f16 --no-cpu-baseline --e2e-chunks 0 --no-split-half > $OUT/bench_f16_tile$t.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/r03_f16/bench_f16_tile$t.log') if x.startswith('{"metric"')][0]
d=json.loads(l)
print('tile cfg $t:', d['value'], 'fps')
for k,v in d['op_breakdown'].items(): print('   ', k, v)
PY
done
