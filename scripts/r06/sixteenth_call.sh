#!/bin/bash
# round 6, sixteenth call: fused window attention with two LDS stages (one barrier per key tile), both arithmetics
# the per-op times of one 68-frame batch (compare profiles/r06_narrow_gemm_ab.log)
OUT=gpurun_out/r06_sixteenth; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_raft.py tests/test_gpu_rfc.py tests/test_gpu_pp.py tests/test_gpu_flow_split.py tests/test_gpu_weight_sweep.py -q -x 2>&1 | tail -6) > $OUT/pytest_flow.log; cat $OUT/pytest_flow.log
python - 4 f32 <<'PY' > $OUT/ops_f32.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "scripts")
import bench_configs as bc
from vsr_amd import engine as E
r = bc.run_propainter(sys.argv[1], sys.argv[2])
print({k: r[k] for k in ("value", "s_per_batch")})
print({k: (v.get("s"), v.get("non_gemm_kernel_ms")) for k, v in r["stages"].items()})
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
head -3 $OUT/ops_f32.log; grep -v "gemm\[" $OUT/ops_f32.log | head -40
python - 4h f16 <<'PY' > $OUT/ops_f16.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "scripts")
import bench_configs as bc
from vsr_amd import engine as E
r = bc.run_propainter(sys.argv[1], sys.argv[2])
print({k: r[k] for k in ("value", "s_per_batch", "psnr_db_vs_exact_mode")})
print({k: (v.get("s"), v.get("non_gemm_kernel_ms")) for k, v in r["stages"].items()})
for k in E.flow_timing_keys():
    if "flash" in k: print(k, E.flow_timing_get(k))
PY
cat $OUT/ops_f16.log | grep -v amdgpu; grep flash $OUT/ops_f32.log
