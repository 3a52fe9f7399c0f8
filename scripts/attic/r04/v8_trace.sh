#!/bin/bash
# per-launch durations of the conv kernels inside the engine, single lane: the 288 x 256 kernel (default) and the 128 x 64 kernel
# (VSR_F32_V8=0), grouped by grid size (= window size T); rocprofv3 kernel trace of two bench steps each
OUT=gpurun_out/r04_v8t; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 0"
for v in 1 0; do
  VSR_F32_V8=$v rocprofv3 --kernel-trace --output-format csv -d $OUT/t$v -o r -- $B > $OUT/trace_$v.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/t$v/**/r_kernel_trace.csv", recursive=True)[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "gather_gemm_f32_v8" in k or "gather_gemm_f32_v3<128" in k:
        g[(k.split("(")[0][:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Workgroup_Size_X", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("V8=$v")
for k, v in sorted(g.items()):
    v.sort()
    print("  %-42s grid %-8s n %-5d  median %8.1f us  mean %8.1f  min %8.1f  max %8.1f" % (k[0], k[1], len(v), v[len(v)//2], sum(v)/len(v), v[0], v[-1]))
PY
done > $OUT/summary.log 2>&1
cat $OUT/summary.log; rm -rf $OUT/t1 $OUT/t0
