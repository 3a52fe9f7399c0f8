#!/bin/bash
# round 6, seventh call: GELU inside the fold + float4 unfold, delta-clean workspace -- parity, then config 4 / 4h
OUT=gpurun_out/r06_seventh; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pp.py tests/test_gpu_golden_wrappers.py -x -q -m gpu 2>&1 | tail -4 > $OUT/pytest_pp.log; cat $OUT/pytest_pp.log
timeout 1500 python -m pytest tests/test_gpu_zbaseline.py -x -q -m gpu -s -k "config4 and default" 2>&1 | grep -E "dB|passed|failed" | tail -6 > $OUT/pytest_config4.log; cat $OUT/pytest_config4.log
for sw in "VSR_PP_FOLD_GELU=0" "VSR_PP_FOLD_GELU=1"; do
  for leg in 4 4h; do
    env $sw timeout 600 python scripts/bench_configs.py $leg 2>/dev/null | grep '^{' > $OUT/cfg_${leg}_$sw.json
    python - <<PY
import json
d = json.loads(open("$OUT/cfg_${leg}_$sw.json").read().splitlines()[-1])
g = d["stages"]["generator"]
print("$sw leg $leg:", d["value"], "fps", d["s_per_batch"], "s/batch psnr", d.get("psnr_db_vs_exact_mode"), "| generator", g["s"], "s non-gemm", g["non_gemm_kernel_ms"], "ms")
PY
  done
done
