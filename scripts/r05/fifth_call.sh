#!/bin/bash
# Round 5, fifth GPU call: the generator after (a) clearing only what a plan addresses when the window geometry changes (was: every
# buffer's capacity, 300 ms per 68-frame batch) and (b) tap-major patch vectors for fold / unfold (k_pp_fold read single floats 196 bytes
# apart: 0.4 TB/s).  Parity first (engine tests, the L=68 fixture), then config 4 in both arithmetics, then kernel stats.
OUT=gpurun_out/r05_fifth; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_pp.py tests/test_gpu_flow_split.py tests/test_gpu_golden_wrappers.py -q -k "not sttn" 2>&1 | tail -4) > $OUT/pytest_pp.log; cat $OUT/pytest_pp.log
(timeout 900 python -m pytest tests/test_gpu_zbaseline.py -q -s -k "config4 or propainter_batch_L20" 2>&1 | grep -E "PSNR|passed|failed|Error|error|skipped" | tail -24) > $OUT/pytest_config4.log; cat $OUT/pytest_config4.log
(timeout 900 python scripts/bench_configs.py 4 4h 4s 2>&1 | grep '^{') > $OUT/configs_pp.log
python - <<'PY'
import json
for line in open("gpurun_out/r05_fifth/configs_pp.log"):
    d = json.loads(line)
    if "error" in d:
        print(d); continue
    print(d["config"], "|", d["value"], "fps", d["s_per_batch"], "s/batch; PSNR vs exact", d["psnr_db_vs_exact_mode"], "fallbacks", d["range_guard_fallbacks"])
    for k, v in d["stages"].items():
        r = v.get("roofline") or {}
        print("   ", k, v.get("s"), "s", v.get("tflops"), "TF", v.get("frac_of_peak"), "| non-GEMM ms", v.get("non_gemm_kernel_ms"), "| dominant", r.get("kernel"), r.get("achieved"), r.get("frac"), "share", r.get("share_of_gemm_time"))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pp -o r -- python scripts/bench_configs.py 4 > $OUT/pp_f32_rocprof.log 2>&1
find $OUT/trace_pp -name "*kernel_stats.csv" -exec cp {} $OUT/propainter_f32_kernel_stats.csv \; ; rm -rf $OUT/trace_pp; head -12 $OUT/propainter_f32_kernel_stats.csv | cut -c1-160
