#!/bin/bash
# round 6, end of round, part A: HBM traffic of every config on the final library (scripts/pmc_configs.py) -- bench.py reads the result
OUT=gpurun_out/r06_final; mkdir -p $OUT; export TMPDIR=/tmp
timeout 3000 python scripts/pmc_configs.py --legs 1,2,3,3d,4,4h,5 --out $OUT/config_traffic.json --workdir $OUT/pmc > $OUT/pmc.log 2>&1; cat $OUT/pmc.log | cut -c1-400; rm -rf $OUT/pmc
ls -la $OUT
