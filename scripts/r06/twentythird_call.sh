#!/bin/bash
# round 6, twenty-third call: square-tile threshold follows the arithmetic (512 exact / 128 fp16 operands) -- propainter suites (incl. the
# accuracy guard, which flips the arithmetic around its checks), config 4 / 4h / 4s lines, file to file
OUT=gpurun_out/r06_twentythird; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1800 python -m pytest tests/test_gpu_pp.py tests/test_gpu_weight_sweep.py tests/test_gpu_flow_split.py tests/test_gpu_golden_wrappers.py -q -x 2>&1 | tail -3) > $OUT/pytest.log; cat $OUT/pytest.log
line() { python scripts/bench_configs.py "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'][:50], d['value'], d['s_per_batch'], d.get('psnr_db_vs_exact_mode'), {k: v.get('s') for k, v in d['stages'].items()})
"; }
line 4h 4 4s 4h | tee $OUT/configs.log
CLIP=gpurun_out/e2e_clip_always.y4m
for p in f16 f32; do (timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 600 --always-on --mode propainter --precision $p 2>&1 | grep '"metric"' | cut -c1-330) | tee -a $OUT/e2e.log; done
rm -f $CLIP
