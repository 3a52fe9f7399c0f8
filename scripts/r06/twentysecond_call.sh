#!/bin/bash
# round 6, twenty-second call: threshold of the square tiles in the generator (VSR_PP_WIDE_N = 512 default / 256 / 128), both arithmetics
OUT=gpurun_out/r06_twentysecond; mkdir -p $OUT; export TMPDIR=/tmp
line() { python scripts/bench_configs.py "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'][:50], d['value'], d['s_per_batch'], d.get('psnr_db_vs_exact_mode'), {k: v.get('s') for k, v in d['stages'].items()})
"; }
for n in 512 256 128; do echo "## VSR_PP_WIDE_N=$n"; VSR_PP_WIDE_N=$n line 4h 4 | tee -a $OUT/wide_n.log; done
