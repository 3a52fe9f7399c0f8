#!/bin/bash
# round 6, fifth call: scores from hi + lo halves in the fp16-operand modes -- the sweep again, and what it costs configs 5 and 4h
OUT=gpurun_out/r06_fifth; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_weight_sweep.py -q -m gpu -s 2>&1 | grep -E "(sttn|generator|flow completion|raft) \[|passed|failed|FAILED" | sed 's/^[.F]*//' | grep -v "print(" > $OUT/pytest_sweep.log; cat $OUT/pytest_sweep.log
for q in 0 1; do
  VSR_F16_QK_SPLIT=$q timeout 600 python scripts/bench_configs.py 5 2>/dev/null | grep '^{' > $OUT/cfg_5_qksplit$q.json
  python - <<PY
import json
d = json.loads(open("$OUT/cfg_5_qksplit$q.json").read().splitlines()[-1])
print("VSR_F16_QK_SPLIT=$q config 5:", d["value"], "fps", d["ms_per_chunk"], "ms/chunk", d["model_tflops"], "TF |", d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"])
PY
done
timeout 600 python scripts/bench_configs.py 4h 2>/dev/null | grep '^{' > $OUT/cfg_4h.json
python - <<PY
import json
d = json.loads(open("$OUT/cfg_4h.json").read().splitlines()[-1])
g = d["stages"]["generator"]
print("leg 4h:", d["value"], "fps", d["s_per_batch"], "s/batch psnr", d.get("psnr_db_vs_exact_mode"), "| generator", g["s"], "s")
PY
timeout 900 python -m pytest tests/test_gpu_zbaseline.py -q -m gpu -s -k "auto_4k" 2>&1 | grep -E "dB|passed|failed" | tail -6
