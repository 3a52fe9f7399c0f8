#!/bin/bash
# round 6, first call: (a) the two-rank dry run of bench.py with the new N-rank legs of configs 5 and 4 (one GPU, gloo),
# (b) HBM traffic of every configuration from PMC passes (scripts/pmc_configs.py)
OUT=gpurun_out/r06_first; mkdir -p $OUT; export TMPDIR=/tmp
VSR_BENCH_DRYRUN_1GPU=1 VSR_BENCH_MULTI_PP_FRAMES=20 VSR_PP_LANES=1 VSR_RAFT_LANES=1 VSR_BENCH_LEG_TIMEOUT=240 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/dryrun_2ranks.log 2>&1
echo "dryrun rc=$?"; grep '^{' $OUT/dryrun_2ranks.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('headline', d['value'], 'n_gpus', d['n_gpus'], 'selftest', (d.get('replicas') or {}).get('selftest', {}).get('ok'))
    for k, v in (d.get('configs_multi') or {}).items():
        print('  multi', k, {a: v.get(a) for a in ('n_ranks', 'value', 'efficiency', 'selftest_ok', 'frames_written', 'error', 'leg_seconds', 'hbm_gbps')})
"
tail -5 $OUT/dryrun_2ranks.log | cut -c1-400
timeout 2400 python scripts/pmc_configs.py --legs 1,5,2,3,3d,4h,4 --out $OUT/config_traffic.json --workdir $OUT/pmc > $OUT/pmc.log 2>&1
echo "pmc rc=$?"; cat $OUT/pmc.log | cut -c1-600
rm -rf $OUT/pmc
