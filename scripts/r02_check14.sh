#!/bin/bash
# y4m colour conversion on the GPU (tests + CLI end-to-end before / after), and the library without the peek in v3/v4/v5
mkdir -p gpurun_out/r02o; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_io.py tests/test_gpu_sttn.py -m gpu -q --tb=short -x --durations=5 2>&1 | tail -25) > gpurun_out/r02o/pytest.log 2>&1
tail -3 gpurun_out/r02o/pytest.log
for c in device host; do
  timeout 600 python scripts/bench_cli.py --frames 300 --color $c > gpurun_out/r02o/cli_$c.log 2>&1
  tail -1 gpurun_out/r02o/cli_$c.log | cut -c1-600
done
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 4 > gpurun_out/r02o/bench.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r02o/bench.log'):
    if l.startswith('{"metric"'):
        d=json.loads(l); print('bench fps',d['value'],'roofline',d['roofline']['achieved'],d['roofline']['frac'],'pcie',d.get('pcie_inclusive',{}).get('value'))
PY
