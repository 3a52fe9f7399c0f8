#!/bin/bash
# round 6, nineteenth call: backward correlation volumes as transposes of the forward ones (VSR_RAFT_CORR_TRANSPOSE, default on)
OUT=gpurun_out/r06_nineteenth; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_raft.py tests/test_gpu_pp.py tests/test_gpu_weight_sweep.py -q -x 2>&1 | tail -3) > $OUT/pytest.log; cat $OUT/pytest.log
for h in 0 1; do echo "## VSR_RAFT_CORR_TRANSPOSE=$h" >> $OUT/raft_ab.log; VSR_RAFT_CORR_TRANSPOSE=$h python scripts/bench_raft.py 2>/dev/null | grep '^{' | cut -c1-400 >> $OUT/raft_ab.log; done; cat $OUT/raft_ab.log
python - 4 f32 <<'PY' > $OUT/ops_f32.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "scripts")
import bench_configs as bc
from vsr_amd import engine as E
r = bc.run_propainter(sys.argv[1], sys.argv[2])
print({k: r[k] for k in ("value", "s_per_batch")})
print({k: (v.get("s"), v.get("tflop"), v.get("non_gemm_kernel_ms")) for k, v in r["stages"].items()})
agg = {}
for k in E.flow_timing_keys():
    ms, n, fl = E.flow_timing_get(k)
    parts = k.split(":")
    eng, kind = parts[:2]
    tag = parts[-1] if kind == "op" else "gemm[" + ":".join(parts[2:5]) + "]:" + parts[-1]
    a = agg.setdefault((eng, tag), [0.0, 0, 0.0]); a[0] += ms; a[1] += n; a[2] += fl
for (eng, tag), (ms, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{eng:5s} {tag:40s} {ms:9.2f} ms {n:6d} launches {fl / ms / 1e9 if ms > 0 and fl > 0 else 0:8.1f} TF")
PY
grep -v amdgpu $OUT/ops_f32.log | head -3; grep "^raft" $OUT/ops_f32.log | head -40
