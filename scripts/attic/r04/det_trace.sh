#!/bin/bash
# one forward of the server detector program (8 frames of 1080p -> 544x960) launch by launch: rocprofv3 kernel trace of
# scripts/r03/det_trace.py, the last forward's launches in order with duration and grid, and totals per kernel
OUT=gpurun_out/r04_det; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o r -- python scripts/r03/det_trace.py > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
python - <<PY > $OUT/launches.log 2>&1
import csv, glob, collections
f = glob.glob("$OUT/t/**/r_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the forwards are identical: find the period = launches between two k_det_normalize calls at the end
idx = [i for i, r in enumerate(rows) if "k_det_normalize" in r["Kernel_Name"]]
last = rows[idx[-8]:] if len(idx) >= 8 else rows
t0 = int(last[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in last)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
print("last forward: %d launches, span %.2f ms, sum of kernel durations %.2f ms" % (len(last), (t1 - t0) / 1e6, busy / 1e6))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in last:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-46s n %-4d %9.1f us  %5.1f %%" % (k, v[0], v[1], 100 * v[1] / (busy / 1e3)))
print("launches over 100 us, in order:")
for r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d >= 100:
        print("  %9.1f us  grid %-8s wg %-4s %s" % (d, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", ""), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]))
PY
cat $OUT/launches.log; rm -rf $OUT/t
