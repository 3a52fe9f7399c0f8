#!/usr/bin/env python3
"""Text detector forward at the 1080p net input (960 x 544): the recorded op-by-op walk on NCHW tensors (VSR_DET_NHWC=0) against the compiled
NHWC-resident plan (ocr_det_nhwc.py), interleaved on one box, net input already on the device.  DET_AB_ONLY=plan|walk runs one side (for
rocprofv3 --kernel-trace --stats)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vsr_amd  # noqa: E402,F401
from vsr_amd.synth import make_det_weights as synthetic_weights  # noqa: E402
from vsr_amd.backend.tools import ocr_det  # noqa: E402
from vsr_amd.backend.tools.paddle_graph import load_graph  # noqa: E402

only = os.environ.get("DET_AB_ONLY", "")
reps = int(os.environ.get("DET_AB_REPS", "10"))
cases = [("ppocr_det_graph.json", 8), ("ppocr_det_graph.json", 16), ("ppocr_det_fast_graph.json", 8)]
if os.environ.get("DET_AB_CASES"):
    cases = [(c.split(":")[0], int(c.split(":")[1])) for c in os.environ["DET_AB_CASES"].split(",")]
for fx, nb in cases:
    g = load_graph(os.path.join(ROOT, "tests", "golden", fx))
    w = synthetic_weights(g)
    x = torch.from_numpy(np.random.default_rng(5).standard_normal((nb, 3, 544, 960)).astype(np.float32)).cuda()
    runners = {}
    for mode, name in (("0", "walk"), ("1", "plan")):
        if only and only != name:
            continue
        r = ocr_det.PaddleGraphRunner(g, w, device=0)
        r.nhwc = mode
        t0 = time.perf_counter()
        for _ in range(3):
            out = r.run_taped(x)
        torch.cuda.synchronize()
        runners[name] = (r, out.clone(), time.perf_counter() - t0)
    if len(runners) == 2:
        d = (runners["plan"][1] - runners["walk"][1]).abs().max().item()
        print(f"{fx} x{nb}: plan vs walk max |diff| of the probability maps {d:.2e}", flush=True)
    res = {}
    for rnd in range(3):
        for name, (r, _, _) in runners.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                r.run_taped(x)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / reps / nb)
    for name, (r, _, setup) in runners.items():
        fl = r.flops[tuple(x.shape)] / nb
        ms = min(res[name])
        extra = ""
        if name == "plan":
            st = r.plan_for(x.shape)
            extra = f"; steps {st['kinds']}; buffers {sum(b.numel() for b in st['bufs'].values()) * 4 / 1e9:.2f} GB"
        print(f"{fx} x{nb} {name}: {ms:.3f} ms/frame (rounds {', '.join('%.3f' % v for v in res[name])}) = {fl / ms / 1e9:.1f} TFLOP/s of {fl / 1e9:.1f} GFLOP/frame "
              f"= {fl / ms / 1e9 / 157.3:.3f} of the fp32 matrix peak; first three calls incl. compile / record {setup:.1f} s{extra}", flush=True)
        r.close()

if os.environ.get("DET_AB_STEPS"):
    # per-step GPU time of the server program's plan (each step 5 x back to back between two events), largest first
    fx, nb = "ppocr_det_graph.json", int(os.environ["DET_AB_STEPS"])
    g = load_graph(os.path.join(ROOT, "tests", "golden", fx))
    r = ocr_det.PaddleGraphRunner(g, synthetic_weights(g), device=0)
    r.nhwc = "1"
    x = torch.from_numpy(np.random.default_rng(5).standard_normal((nb, 3, 544, 960)).astype(np.float32)).cuda()
    for _ in range(2):
        r.run_taped(x)
    st = r.plan_for(x.shape)
    rows = []
    for (kind, p), (fn, args) in zip(st["launches"], st["tape"]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(*args)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        if kind == "gemm":
            fl = sum(2.0 * q["M"] * q["N"] * q["K"] for q in p)
            q = p[-1]
            rows.append((ms, f"{'+'.join(q_['tag'] for q_ in p):<10} M {q['M']:>8} N {q['N']:>5} K {q['K']:>6} tile {({0: '128x128', 1: '256x32', 3: '128x64'})[q['tile_cfg']]} "
                             f"v{q['variant']} R {int(q['R'] is not None)} padded {fl / ms / 1e9:6.1f} TF"))
        else:
            rows.append((ms, f"{kind} {p.get('C', '')} {p.get('Ho', p.get('H', ''))}x{p.get('Wo', p.get('W', ''))}"))
    tot = sum(m for m, _ in rows)
    print(f"per-step times, {nb} frames: sum {tot:.2f} ms = {tot / nb:.3f} ms/frame over {len(rows)} steps")
    for ms, txt in sorted(rows, key=lambda t: -t[0])[:70]:
        print(f"  {ms:7.3f} ms  {100 * ms / tot:5.1f} %  {txt}")
    r.close()
