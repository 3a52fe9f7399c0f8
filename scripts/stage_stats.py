#!/usr/bin/env python3
"""Per-ENGINE kernel summaries of one propainter batch (BASELINE config 4), from ONE rocprofv3 kernel trace.

    python scripts/stage_stats.py --leg 4|4h [--out profiles] [--tag r06]

The three engines of the plugin (RAFT, flow completion, generator) launch the same gather-GEMM symbols, so a `--stats` summary of the
whole batch cannot say at what rate any of them runs.  This runs `rocprofv3 --kernel-trace` over `scripts/bench_configs.py --stages <leg>`:
one warm 68-frame call, then one call in the plugin's profile mode (every stage on one stream, device-synchronised between stages) in
which a marker kernel closes every stage (fmod = begin, nextafter = raft, hypot = flow completion, copysign = other, logaddexp =
generator).  The dispatches between two markers are the stage's; per stage one CSV in the columns of rocprofv3's own kernel stats
(Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs) -> <out>/<tag>_propainter_<f32|f16>_<stage>_kernel_stats.csv.
"""
import argparse
import csv
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARKS = (("fmod", "begin"), ("nextafter", "raft"), ("hypot", "flow_completion"), ("copysign", "other"), ("logaddexp", "generator"))


def stage_of_marker(name):
    for key, stage in MARKS:
        if key in name:
            return stage
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", default="4")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles"))
    ap.add_argument("--tag", default="r06")
    args = ap.parse_args()
    work = os.path.join(ROOT, "gpurun_out", "stage_stats_" + args.leg)
    shutil.rmtree(work, ignore_errors=True)
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", work, "-o", "r", "--", sys.executable,
           os.path.join(ROOT, "scripts", "bench_configs.py"), "--stages", args.leg]
    r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=2400)
    path = None
    for base, _, files in os.walk(work):
        for f in files:
            if f.endswith("kernel_trace.csv"):
                path = os.path.join(base, f)
    if path is None:
        raise SystemExit(f"no kernel trace: rc {r.returncode}\n{r.stdout[-800:]}\n{r.stderr[-800:]}")
    rows = list(csv.DictReader(open(path, newline="")))
    rows.sort(key=lambda x: int(x["Start_Timestamp"]))
    stages = defaultdict(lambda: defaultdict(list))
    cur, started = [], False
    for x in rows:
        st = stage_of_marker(x["Kernel_Name"])
        if st is None:
            cur.append(x)
            continue
        if st == "begin":
            started, cur = True, []
            continue
        if started:
            for y in cur:
                stages[st][y["Kernel_Name"]].append(int(y["End_Timestamp"]) - int(y["Start_Timestamp"]))
        cur = []
    os.makedirs(args.out, exist_ok=True)
    prec = {"4": "f32", "4h": "f16", "4s": "f16_raft_split"}.get(args.leg, args.leg)
    for st, kern in stages.items():
        tot = sum(sum(v) for v in kern.values())
        out = os.path.join(args.out, f"{args.tag}_propainter_{prec}_{st}_kernel_stats.csv")
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for name, v in sorted(kern.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / tot, 4) if tot else 0, min(v), max(v)])
        top = sorted(kern.items(), key=lambda kv: -sum(kv[1]))[:3]
        print(f"{st}: {tot / 1e6:.1f} ms of kernels in {sum(len(v) for v in kern.values())} launches; " +
              "; ".join(f"{n.split('(')[0][:60]} {sum(v) / 1e6:.1f} ms" for n, v in top), flush=True)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
