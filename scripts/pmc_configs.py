#!/usr/bin/env python3
"""HBM traffic of every BASELINE configuration, per kernel symbol and per unit of work, from rocprofv3 PMC passes.

    python scripts/pmc_configs.py [--legs 2,3,3d,4,4h,5,1] [--out profiles/config_traffic.json] [--tag r06]

For each leg two SEPARATE rocprofv3 runs (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC
slots") of `python scripts/bench_configs.py --unit <leg>`, which runs one warm unit, a marker kernel, ONE measured unit (a 50-frame
chunk / a 47-frame sttn-det batch / an 8-frame detector forward (VSR_DET_BATCH, what the detector pass hands over) / a 68-frame propainter batch) and a marker kernel again.  Only the
dispatches between the two markers are counted.  Leg "1" is the headline (1080p sttn-auto chunk).

Correction (MI355X_MICROARCH.md, HBM): on gfx950 FETCH_SIZE reports half the bytes of a wide streaming read, WRITE_SIZE is exact
(both re-checked here on the __amd_rocclr_copyBuffer dispatches of the same run when there are any):
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024        (the counters are in KB)

Output (JSON): {"legs": {leg: {"hbm_bytes_per_unit", "unit", "kernels": {symbol: {"launches", "hbm_bytes_per_launch",
"fetch_kb_mean", "write_kb_mean"}}, "note"}}}; bench_configs.attach_traffic() reads it back into each config's `roofline.traffic`,
`hbm_gb_per_unit` and `hbm_gbps`.  No --kernel-trace / --stats in a counter run (the pool refuses the combination).
"""
import argparse
import csv
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = {"1": "one 50-frame 1080p sttn-auto chunk", "2": "one 50-frame 720p sttn-auto chunk", "3": "one 47-frame 1080p sttn-det batch",
         "3d": "one 8-frame detector forward (960x544)", "4": "one 68-frame 1080p propainter batch (exact fp32)",
         "4h": "one 68-frame 1080p propainter batch (reference GPU arithmetic)", "4s": "one 68-frame propainter batch (f16 + split RAFT)",
         "5": "one 50-frame 4K sttn-auto chunk (fp16 operands)", "5x": "one 50-frame 4K sttn-auto chunk (fp32)"}
MARK = "nextafter"


def symbol(kernel_name):
    """'void gather_gemm_f32_v3<128, 64, 2, 2, 0>(GGProblem const*, ...)' -> 'gather_gemm_f32_v3<128, 64, 2, 2, 0>'"""
    s = kernel_name.strip()
    s = re.sub(r"^void\s+", "", s)
    depth = 0
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return s[:i]
    return s


def collect(leg, counter, workdir):
    d = os.path.join(workdir, f"{leg}_{counter}")
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "r", "--", sys.executable,
           os.path.join(ROOT, "scripts", "bench_configs.py"), "--unit", leg]
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    path = None
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith("counter_collection.csv"):
                path = os.path.join(base, f)
    if path is None:
        raise RuntimeError(f"no counter_collection.csv for leg {leg} {counter}: rc {r.returncode}\n{r.stdout[-600:]}\n{r.stderr[-600:]}")
    rows = list(csv.DictReader(open(path, newline="")))
    rows.sort(key=lambda x: int(x["Dispatch_Id"]))
    marks = [i for i, x in enumerate(rows) if MARK in x["Kernel_Name"] and x["Counter_Name"] == counter]
    if len(marks) < 2:
        raise RuntimeError(f"leg {leg} {counter}: {len(marks)} marker dispatches (need 2)")
    inside = [x for x in rows[marks[-2] + 1:marks[-1]] if x["Counter_Name"] == counter]
    agg = defaultdict(lambda: [0, 0.0])
    for x in inside:
        k = symbol(x["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(x["Counter_Value"])
    cal = [float(x["Counter_Value"]) for x in rows if "copyBuffer" in x["Kernel_Name"] and x["Counter_Name"] == counter]
    shutil.rmtree(d, ignore_errors=True)
    return agg, cal


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--legs", default="1,2,3,3d,4,4h,5")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "config_traffic.json"))
    ap.add_argument("--tag", default="r06")
    ap.add_argument("--workdir", default=os.path.join(ROOT, "gpurun_out", "pmc_configs"))
    args = ap.parse_args()
    os.makedirs(args.workdir, exist_ok=True)
    try:
        result = json.load(open(args.out))
    except (OSError, ValueError):
        result = {"legs": {}}
    result["tag"] = args.tag
    result["correction"] = "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of wide reads, MI355X_MICROARCH.md HBM section)"
    for leg in args.legs.split(","):
        try:
            fa, fcal = collect(leg, "FETCH_SIZE", args.workdir)
            wa, wcal = collect(leg, "WRITE_SIZE", args.workdir)
        except Exception as e:      # noqa: BLE001 -- one leg failing leaves the others
            print(f"leg {leg}: FAILED {e!r}"[:500], flush=True)
            continue
        kernels, total = {}, 0.0
        for k in sorted(set(fa) | set(wa)):
            nf, tf = fa.get(k, [0, 0.0])
            nw, tw = wa.get(k, [0, 0.0])
            n = max(nf, nw)
            b = (2 * tf + tw) * 1024
            total += b
            kernels[k] = {"launches": n, "hbm_bytes_per_launch": int(b / n) if n else 0, "fetch_kb_mean": round(tf / nf, 1) if nf else 0.0,
                          "write_kb_mean": round(tw / nw, 1) if nw else 0.0, "hbm_bytes_total": int(b)}
        result["legs"][leg] = {"unit": UNITS.get(leg, leg), "hbm_bytes_per_unit": int(total), "kernels": kernels,
                               "calibration_copyBuffer_kb": {"FETCH_SIZE_mean": round(sum(fcal) / len(fcal), 1) if fcal else None,
                                                             "WRITE_SIZE_mean": round(sum(wcal) / len(wcal), 1) if wcal else None},
                               "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) over the dispatches of {UNITS.get(leg, leg)}; "
                                       "bytes = (2*FETCH + WRITE)*1024"}
        top = sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_total"])[:4]
        print(f"leg {leg}: {total / 1e9:.2f} GB per unit; " + "; ".join(f"{k} x{v['launches']} {v['hbm_bytes_per_launch'] / 1e6:.1f} MB" for k, v in top), flush=True)
        json.dump(result, open(args.out, "w"), indent=1)
    tagged = os.path.join(os.path.dirname(args.out), f"{args.tag}_config_traffic.json")
    if os.path.abspath(tagged) != os.path.abspath(args.out):
        shutil.copy(args.out, tagged)


if __name__ == "__main__":
    main()
