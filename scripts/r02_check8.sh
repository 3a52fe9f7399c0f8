#!/bin/bash
mkdir -p gpurun_out/r02h
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "softmax" > gpurun_out/r02h/pytest_softmax.log 2>&1
tail -2 gpurun_out/r02h/pytest_softmax.log
timeout 900 python -m pytest tests/test_gpu_sttn.py tests/test_gpu_pp.py -m gpu -q -x > gpurun_out/r02h/pytest_sttn_pp.log 2>&1
tail -2 gpurun_out/r02h/pytest_sttn_pp.log
timeout 600 python bench.py --no-cpu-baseline --e2e-chunks 0 > gpurun_out/r02h/bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r02h/bench.log") if x.startswith('{')]
d=json.loads(l[-1]); print(d["value"], d["ms_per_step"], {k:d[k]["value"] for k in ("split_half_mode","split_format_mode","fp16_mode")})
print({k:(v['ms'],v['launches']) for k,v in d["op_breakdown"].items()})
PY
