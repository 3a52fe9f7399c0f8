#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (written by scripts/profile_round.sh on the GPU box) into profiles/<tag>_*.

  <tag>_kernel_stats.csv       rocprofv3 --kernel-trace --stats per-kernel summary of the bench command
  <tag>_pmc_summary.csv        per-kernel mean / total of every PMC counter collected (separate passes)
  <tag>_dominant_kernel_pmc.json + dominant_kernel_pmc.json (read by bench.py for roofline.traffic)
  <tag>_bench.log, _pytest_gpu.log, _smoke.log
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)

for name in ("bench.log", "pytest_gpu.log", "smoke.log", "bench_under_rocprof.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
stats = os.path.join(src, "trace", "r_kernel_stats.csv")
if os.path.exists(stats):
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))

# ---- PMC: counter_collection rows are per dispatch and per counter
agg = defaultdict(lambda: [0, 0.0])
for d in sorted(os.listdir(src)):
    p = os.path.join(src, d, "r_counter_collection.csv")
    if not d.startswith("pmc") or not os.path.exists(p):
        continue
    with open(p, newline="") as f:
        for row in csv.DictReader(f):
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "Counter_Name", "dispatches", "mean", "total"])
    for (k, c), (n, tot) in sorted(agg.items()):
        w.writerow([k, c, n, tot / n, tot])

# ---- the fp16-operand mode's own passes (f16_trace, f16_pmc_*)
p16 = os.path.join(src, "f16_trace", "r_kernel_stats.csv")
if os.path.exists(p16):
    shutil.copy(p16, os.path.join(dst, f"{tag}_f16_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "f16_bench_under_rocprof.json")):
    shutil.copy(os.path.join(src, "f16_bench_under_rocprof.json"), os.path.join(dst, f"{tag}_f16_bench_under_rocprof.json"))
agg16 = defaultdict(lambda: [0, 0.0])
for d in sorted(os.listdir(src)):
    p = os.path.join(src, d, "r_counter_collection.csv")
    if not d.startswith("f16_pmc") or not os.path.exists(p):
        continue
    with open(p, newline="") as f:
        for row in csv.DictReader(f):
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            agg16[k][0] += 1
            agg16[k][1] += float(row["Counter_Value"])
if agg16:
    with open(os.path.join(dst, f"{tag}_f16_pmc_summary.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "Counter_Name", "dispatches", "mean", "total"])
        for (k, c), (n, tot) in sorted(agg16.items()):
            w.writerow([k, c, n, tot / n, tot])

# ---- dominant kernel: largest total duration in the kernel-trace stats
dom = None
if os.path.exists(stats):
    with open(stats, newline="") as f:
        rows = list(csv.DictReader(f))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    dom = rows[0]["Name"].split("(")[0]
    print("dominant:", dom, "avg ms", float(rows[0]["AverageNs"]) / 1e6, "calls", rows[0]["Calls"])
if dom and (dom, "FETCH_SIZE") in agg and (dom, "WRITE_SIZE") in agg:
    nf, tf = agg[(dom, "FETCH_SIZE")]
    nw, tw = agg[(dom, "WRITE_SIZE")]
    cal = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if ("__amd_rocclr_copyBuffer", c) in agg:
            n, t = agg[("__amd_rocclr_copyBuffer", c)]
            cal[c + "_KB_mean_copyBuffer"] = t / n
    out = {"kernel": dom, "launches": nf, "FETCH_SIZE_KB_mean": tf / nf, "WRITE_SIZE_KB_mean": tw / nw,
           "hbm_bytes_per_launch": int(round((2 * tf / nf + tw / nw) * 1024)), "calibration": cal,
           "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python bench.py --steps 1 --warmup 0 "
                   "--no-cpu-baseline --no-split-half --e2e-chunks 0`; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 1/2 of "
                   "wide (16 B/lane) reads, WRITE_SIZE is exact KB (both checked on __amd_rocclr_copyBuffer in the same run, see "
                   "calibration), so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; mean over all launches of the symbol"}
    for name in (f"{tag}_dominant_kernel_pmc.json", "dominant_kernel_pmc.json"):
        with open(os.path.join(dst, name), "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out)[:300])
