#!/bin/bash
# round 6, twenty-first call: wide GEMMs of the generator on 128 x 128 tiles (default now) -- propainter suites, config 4 / 4h lines, 3 generator lanes
OUT=gpurun_out/r06_twentyfirst; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_pp.py tests/test_gpu_weight_sweep.py tests/test_gpu_flow_split.py -q -x 2>&1 | tail -3) > $OUT/pytest.log; cat $OUT/pytest.log
line() { python scripts/bench_configs.py "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'][:50], d['value'], d['s_per_batch'], d.get('psnr_db_vs_exact_mode'), {k: v.get('s') for k, v in d['stages'].items()})
"; }
echo "## default"; line 4 4h | tee $OUT/default.log
echo "## VSR_PP_LANES=3"; VSR_PP_LANES=3 line 4h | tee $OUT/lanes3.log
echo "## VSR_PP_TR_TILE=128x64"; VSR_PP_TR_TILE=128x64 line 4h | tee $OUT/tile64.log
