#!/bin/bash
# round 6, sixth call: the accuracy guard -- the sweep again, the GPU suite around the engines, config 5 / 4h
OUT=gpurun_out/r06_sixth; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_weight_sweep.py -q -m gpu -s 2>&1 | grep -E "(sttn|generator|flow completion|raft) \[|passed|failed|FAILED" | sed 's/^[.F]*//' | grep -v "print(" > $OUT/pytest_sweep.log; cat $OUT/pytest_sweep.log
timeout 2400 python -m pytest tests/test_gpu_sttn.py tests/test_gpu_flow_split.py tests/test_gpu_pp.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -6 > $OUT/pytest_engines.log; cat $OUT/pytest_engines.log
for leg in 5 4h; do
  timeout 600 python scripts/bench_configs.py $leg 2>/dev/null | grep '^{' > $OUT/cfg_$leg.json
  python - <<PY
import json
d = json.loads(open("$OUT/cfg_$leg.json").read().splitlines()[-1])
print("config $leg:", d["value"], "fps", d.get("ms_per_chunk") or d.get("s_per_batch"), d.get("fp32_fallback_chunks", d.get("range_guard_fallbacks")), d.get("psnr_db_vs_exact_mode"))
PY
done
