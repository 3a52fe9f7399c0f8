#!/bin/bash
mkdir -p gpurun_out/r02d
timeout 300 python scripts/bench_lama.py > gpurun_out/r02d/bench_lama.log 2>&1
timeout 600 python bench.py --precision f16 --no-cpu-baseline --no-split-half --e2e-chunks 0 --res 4k > gpurun_out/r02d/bench_f16_4k.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-split-half --e2e-chunks 4 > gpurun_out/r02d/bench_e2e.log 2>&1
timeout 300 python -m pytest tests/test_gpu_lama.py -m gpu -q -x > gpurun_out/r02d/pytest_lama.log 2>&1
tail -2 gpurun_out/r02d/pytest_lama.log; grep metric gpurun_out/r02d/bench_lama.log | cut -c1-300
python - <<'PY'
import json
for f in ("gpurun_out/r02d/bench_f16_4k.log","gpurun_out/r02d/bench_e2e.log"):
    l=[x for x in open(f) if x.startswith('{')]
    if not l: print(f, "no line"); continue
    d=json.loads(l[-1]); print(f, d["value"], d["ms_per_step"], d.get("pcie_inclusive"))
    for k,v in d["op_breakdown"].items(): print("   ",k,v)
PY
