#!/bin/bash
# the profiled call of bench_configs' config-4 leg after a warm-up of its own (its single-stream run structure built plans inside the timed stage)
OUT=gpurun_out/r05_eleventh; mkdir -p $OUT
(timeout 900 python scripts/bench_configs.py 4 4h 2>&1 | grep '^{') > $OUT/configs_pp.log
python - <<'PY'
import json
for line in open("gpurun_out/r05_eleventh/configs_pp.log"):
    d = json.loads(line)
    print(d.get("config"), "|", d.get("value"), "fps", d.get("s_per_batch"), "s/batch", d.get("roofline_stage"), {k: (v.get("s"), v.get("tflops")) for k, v in d.get("stages", {}).items()}, (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"), d.get("error"), d.get("leg_seconds"))
PY
(timeout 300 python -m pytest tests/test_gpu_pp.py -q -k "lanes or plugin_matches" 2>&1 | tail -2)
