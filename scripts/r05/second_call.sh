#!/bin/bash
# Round 5, second GPU call: the column-range tests again (the first call's failure was the test's own parametrisation), the fp16-operand
# mode of the flow engines (kernel variant 7), the new BASELINE-size parity cases (portrait sttn-det, 4K fp32), config 4 at its real batch
# size in three arithmetics with per-stage / per-kernel timing, bench.py's `configs` object and its self-launch with --gpus 2.
OUT=gpurun_out/r05_second; mkdir -p $OUT
(VSR_DECODE_COLS=1 timeout 600 python -m pytest tests/test_gpu_sttn.py -q -k "decoder_box" 2>&1 | tail -8) > $OUT/pytest_cols.log; tail -2 $OUT/pytest_cols.log
(timeout 600 python -m pytest tests/test_gpu_flow_split.py tests/test_gpu_pp.py -q -s 2>&1 | grep -E "vs exact|passed|failed|Error|error" | tail -20) > $OUT/pytest_flow_f16.log; cat $OUT/pytest_flow_f16.log
(timeout 900 python -m pytest tests/test_gpu_zbaseline.py -q -s -k "portrait or auto_4k" 2>&1 | grep -E "PSNR|passed|failed|Error" | tail -12) > $OUT/pytest_zbaseline_new.log; cat $OUT/pytest_zbaseline_new.log
(timeout 900 python scripts/bench_configs.py 4 4h 4s 2>&1 | grep '^{') > $OUT/configs_pp.log; cut -c1-700 $OUT/configs_pp.log
(timeout 600 python bench.py --no-cpu-baseline --no-split-half --no-full-work --e2e-chunks 0 --steps 3 --warmup 1 --configs 2,3,3d,5 2>&1 | grep '"metric"') > $OUT/bench_configs_line.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_second/bench_configs_line.json").read())
print("headline", d["value"], "fps")
for k, v in d.get("configs", {}).items():
    print(k, {a: v.get(a) for a in ("value", "unit", "model_tflops", "error", "leg_seconds")}, (v.get("roofline") or {}).get("kernel"), (v.get("roofline") or {}).get("frac"))
PY
VSR_BENCH_DRYRUN_1GPU=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/selflaunch_2ranks.log 2>&1; echo "self-launch rc=$?"
grep -o '"value": [0-9.]*, "unit": "frames/s", "n_gpus": 2' $OUT/selflaunch_2ranks.log; grep -o '"selftest": {[^}]*}' $OUT/selflaunch_2ranks.log | cut -c1-200
