#!/bin/bash
mkdir -p gpurun_out/r02e
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "fp16_operands or split_format" > gpurun_out/r02e/pytest_kernels.log 2>&1
tail -3 gpurun_out/r02e/pytest_kernels.log
timeout 600 python -m pytest tests/test_gpu_sttn.py -m gpu -q -x -s -k "fp16 or range_guard or split" > gpurun_out/r02e/pytest_sttn.log 2>&1
grep -E "f16|passed|failed" gpurun_out/r02e/pytest_sttn.log | tail -5
B="python bench.py --precision f16 --no-cpu-baseline --no-split-half --e2e-chunks 0"
for st in 3 2 4; do
  VSR_V6_STAGES=$st timeout 300 $B > gpurun_out/r02e/bench_f16_v6_st$st.log 2>&1
done
VSR_F16_KERNEL=5 timeout 300 $B > gpurun_out/r02e/bench_f16_v5.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02e/bench_f16_*.log")):
    l=[x for x in open(f) if x.startswith('{')]
    if not l: print(f, "no line", open(f).read()[-300:]); continue
    d=json.loads(l[-1]); print(f, d["value"], d["ms_per_step"])
    print("   ", {k:(v['ms'],v['tflops']) for k,v in d["op_breakdown"].items()})
PY
