#!/bin/bash
# Round 5, third GPU call: BASELINE config 4 at its real batch size.  Round 4's flow-completion plan did not build beyond 49 frames of
# 1920x360 ("offset table entry exceeds int32": found by call 2) -- fixed by cutting the two large convs into frame groups.
OUT=gpurun_out/r05_third; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_zbaseline.py -q -s -k "config4" 2>&1 | grep -E "PSNR|passed|failed|Error|error|skipped" | tail -24) > $OUT/pytest_config4.log; cat $OUT/pytest_config4.log
(timeout 900 python scripts/bench_configs.py 4 4h 4s 2>&1 | grep '^{') > $OUT/configs_pp.log
python - <<'PY'
import json
for line in open("gpurun_out/r05_third/configs_pp.log"):
    d = json.loads(line)
    if "error" in d:
        print(d); continue
    print(d["config"], "|", d["value"], "fps", d["s_per_batch"], "s/batch; PSNR vs exact", d["psnr_db_vs_exact_mode"], "fallbacks", d["range_guard_fallbacks"])
    for k, v in d["stages"].items():
        r = v.get("roofline") or {}
        print("   ", k, v.get("s"), "s", v.get("tflops"), "TF", v.get("frac_of_peak"), "| non-GEMM ms", v.get("non_gemm_kernel_ms"), "| dominant", r.get("kernel"), r.get("achieved"), r.get("frac"), "share", r.get("share_of_gemm_time"))
PY
(VSR_RFC_TEST=1 timeout 600 python -m pytest tests/test_gpu_rfc.py tests/test_gpu_pp.py tests/test_gpu_raft.py -q 2>&1 | tail -3) > $OUT/pytest_engines.log; cat $OUT/pytest_engines.log
