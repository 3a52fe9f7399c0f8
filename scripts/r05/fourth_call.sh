#!/bin/bash
# Round 5, fourth GPU call: (1) the by-offset file access of tools/rank_io.py on the GPU (one process and two ranks, byte-for-byte files);
# (2) rocprofv3 kernel-trace stats of BASELINE config 4 at its real batch size (fp32 and the reference's GPU arithmetic) and of every
# engine that has a number in DESIGN section 5 (RAFT, flow completion, LaMa, detector): the r05 summaries the rooflines are checked against.
OUT=gpurun_out/r05_fourth; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_io.py tests/test_gpu_multirank.py -q -k "resident_chunk_loop or two_ranks" 2>&1 | tail -6) > $OUT/pytest_by_offset.log; cat $OUT/pytest_by_offset.log
prof() {   # prof NAME CMD...: kernel-trace stats of CMD -> $OUT/NAME_kernel_stats.csv (+ the command's own JSON lines)
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o r -- "$@" > $OUT/$name.log 2>&1
  cp $OUT/trace_$name/r_kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null || find $OUT/trace_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
  rm -rf $OUT/trace_$name
  grep '^{' $OUT/$name.log | cut -c1-400
  head -8 $OUT/${name}_kernel_stats.csv | cut -c1-200
}
prof propainter_f32 python scripts/bench_configs.py 4
prof propainter_f16 python scripts/bench_configs.py 4h
prof raft python scripts/bench_raft.py
prof rfc python scripts/bench_rfc.py
prof lama python scripts/bench_lama.py
prof detector python scripts/bench_configs.py 3d
# per-op time of the generator's and RAFT's non-GEMM kernels (HIP events, one 68-frame batch)
python - <<'PY' > $OUT/pp_nongemm_ops.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "scripts")
import numpy as np, torch
import bench_configs as bc
from vsr_amd import engine as E
r = bc.run_propainter("4", "f32")
agg = {}
for k in E.flow_timing_keys():
    ms, n, fl = E.flow_timing_get(k)
    eng, kind = k.split(":")[:2]
    tag = k.split(":")[-1] if kind == "op" else "gemm:" + k.split(":")[-1]
    a = agg.setdefault((eng, tag), [0.0, 0, 0.0]); a[0] += ms; a[1] += n; a[2] += fl
for (eng, tag), (ms, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{eng:5s} {tag:28s} {ms:9.2f} ms {n:6d} launches {fl / ms / 1e9 if ms > 0 and fl > 0 else 0:8.1f} TF")
PY
head -50 $OUT/pp_nongemm_ops.log
