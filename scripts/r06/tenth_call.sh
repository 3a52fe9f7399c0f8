#!/bin/bash
# round 6, tenth call: the dot-product kernel for GEMMs of <= 4 output columns (gather_gemm_narrow.h) -- kernel tests, the flow engines'
# suites on it, config 4 / 4h with it on and off (per-op HIP-event times of one 68-frame batch); and the detector's HIP-graph replay
# (opt-in, parked since round 1 with a memory access fault) run once more under a timeout to see what it does today.
OUT=gpurun_out/r06_tenth; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "dot_product or narrow" 2>&1 | tail -5) > $OUT/pytest_narrow.log; cat $OUT/pytest_narrow.log
(timeout 1500 python -m pytest tests/test_gpu_raft.py tests/test_gpu_rfc.py tests/test_gpu_pp.py tests/test_gpu_flow_split.py -q -x 2>&1 | tail -6) > $OUT/pytest_flow.log; cat $OUT/pytest_flow.log
ops() {   # ops NAME LEG PRECISION: per-op time of one batch, every op, by time
python - "$2" "$3" <<'PY' > $OUT/$1.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "scripts")
import bench_configs as bc
from vsr_amd import engine as E
r = bc.run_propainter(sys.argv[1], sys.argv[2])
print({k: r[k] for k in ("value", "s_per_batch")})
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
head -4 $OUT/$1.log; grep -E "narrow|\[1:0" $OUT/$1.log
}
VSR_GG_NARROW=0 ops ops_f32_mfma 4 f32
ops ops_f32_narrow 4 f32
VSR_GG_NARROW=0 ops ops_f16_mfma 4h f16
ops ops_f16_narrow 4h f16
# the detector's graph replay
(VSR_DET_GRAPH=1 timeout 300 python -m pytest tests/test_gpu_ocr_det.py -q -x -k graph_replay 2>&1 | tail -25) > $OUT/det_graph.log; echo "det graph rc=$?"; tail -25 $OUT/det_graph.log
