#!/bin/bash
# round 6, eighteenth call: RAFT with the context's share of the GRU convs computed once per call (VSR_RAFT_CTX_HOIST, default on) --
# the RAFT / propainter / weight-sweep suites, RAFT alone on and off, config 4 / 4h lines
OUT=gpurun_out/r06_eighteenth; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_raft.py tests/test_gpu_pp.py tests/test_gpu_weight_sweep.py tests/test_gpu_flow_split.py -q -x 2>&1 | sed 's/^[.F]*//' | grep -i "raft\|passed\|failed\|error" | tail -30) > $OUT/pytest.log; cat $OUT/pytest.log
for h in 0 1; do echo "## VSR_RAFT_CTX_HOIST=$h" >> $OUT/raft_ab.log; VSR_RAFT_CTX_HOIST=$h python scripts/bench_raft.py 2>/dev/null | grep '^{' | cut -c1-700 >> $OUT/raft_ab.log; done; cat $OUT/raft_ab.log
python scripts/bench_configs.py 4 4h > $OUT/configs_4.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r06_eighteenth/configs_4.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["config"][:40], d["value"], d["s_per_batch"], d.get("psnr_db_vs_exact_mode"), {k: (v.get("s"), v.get("tflop"), v.get("non_gemm_kernel_ms")) for k, v in d["stages"].items()})
PY
