OUT=gpurun_out/r06_last; mkdir -p $OUT; export TMPDIR=/tmp
cp profiles/config_traffic.json $OUT/config_traffic.json
timeout 900 python scripts/pmc_configs.py --legs 3d --out $OUT/config_traffic.json --tag r06 --workdir /tmp/pmc_work 2>&1 | grep -v amdgpu.ids | tail -3
cp $OUT/config_traffic.json profiles/config_traffic.json
(timeout 600 python scripts/bench_configs.py 3d 2>/dev/null | grep '^{') > $OUT/cfg_3d.json; cut -c1-900 $OUT/cfg_3d.json
