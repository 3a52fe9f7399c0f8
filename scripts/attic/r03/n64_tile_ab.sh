#!/bin/bash
# tile of the N = 64 convs (decoder / encoder): 256x64 (4x1 waves, 385 VGPRs -> 1 wave per SIMD) vs 128x64 (3 workgroups per CU)
OUT=gpurun_out/r03_n64; mkdir -p $OUT
for t in 2 3; do
  VSR_N64_TILE=$t python bench.py --no-cpu-baseline --e2e-chunks 0 --no-split-half > $OUT/bench_n64_$t.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/r03_n64/bench_n64_$t.log') if x.startswith('{"metric"')]
if not l: print(open('gpurun_out/r03_n64/bench_n64_$t.log').read()[-1500:])
else:
    d=json.loads(l[0]); print('VSR_N64_TILE=$t:', d['value'], 'fps; single lane', d['single_lane']['value'], '; dec', d['op_breakdown']['dec'], 'enc', d['op_breakdown']['enc'])
PY
done
VSR_N64_TILE=3 timeout 300 python -m pytest tests/test_gpu_sttn.py -m gpu -x -q -k "small_windows or default_windows or det_inpaint" 2>&1 | tail -2
