#!/usr/bin/env python3
"""Triage: the compiled detector plan step by step on the GPU against its CPU replay (tests/_det_replay.py) -- the first step whose output
buffer differs is printed with its parameters."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vsr_amd  # noqa: E402,F401
import _det_replay  # noqa: E402
from oracle.ppocr_det import synthetic_weights  # noqa: E402
from vsr_amd.backend.tools import ocr_det  # noqa: E402
from vsr_amd.backend.tools.paddle_graph import load_graph  # noqa: E402

fx = sys.argv[1] if len(sys.argv) > 1 else "ppocr_det_graph.json"
nb, H, W = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (2, 64, 96)))
g = load_graph(os.path.join(ROOT, "tests", "golden", fx))
w = synthetic_weights(g)
r = ocr_det.PaddleGraphRunner(g, w, device=0)
r.nhwc = "1"
x = np.random.default_rng(1).standard_normal((nb, 3, H, W)).astype(np.float32)
st = r.plan_for(x.shape)
plan, tape, bufs = st["plan"], st["tape"], st["bufs"]
assert len(tape) == len(st["launches"])
last_of = {}                                   # index of the plan step that completes tape entry k (a group of GEMMs is one launch)
k = 0
for i, (kind, p) in enumerate(plan.steps):
    nxt = plan.steps[i + 1] if i + 1 < len(plan.steps) else (None, None)
    if kind == "gemm" and p["group"] is not None and nxt[0] == "gemm" and nxt[1]["group"] == p["group"]:
        continue
    last_of[i] = k
    k += 1
assert k == len(tape)
st["xin"].copy_(torch.from_numpy(x).reshape(-1).cuda())
r._sa.value = torch.cuda.current_stream().cuda_stream
bad = []


def after(i, kind, p, cpu):
    if i not in last_of:                       # inside a group: the launch comes with its last member (the members share nothing)
        return
    fn, args = tape[last_of[i]]
    assert fn(*args) == 0
    torch.cuda.synchronize()
    name = p["C"] if kind == "gemm" else p["dst"] if kind == "copy" else p["out"]
    got, want = bufs[name].cpu().numpy()[:cpu[name].size], cpu[name]
    m = np.isfinite(want)
    d = np.abs(got[m] - want[m])
    scale = max(1e-6, float(np.abs(want[m]).max()))
    worst = float(d.max()) if d.size else 0.0
    flag = worst > 2e-4 * scale
    if flag or os.environ.get("VERBOSE"):
        small = {k: v for k, v in p.items() if not isinstance(v, dict)}
        print(f"step {i} {kind}: max |gpu - cpu| {worst:.3e} (scale {scale:.3e}) {'<-- DIFFERS' if flag else ''} {small}", flush=True)
    if flag:
        bad.append(i)
        idx = np.flatnonzero(m)[np.argsort(-d)[:5]]
        print("   worst offsets", idx.tolist(), "gpu", got[idx].tolist(), "cpu", want[idx].tolist(), flush=True)
        bufs[name][:want.size].copy_(torch.from_numpy(np.where(m, want, 0).astype(np.float32)).cuda())      # carry on from the CPU's state


out, _ = _det_replay.run_plan(plan, x, after_step=after)
print(f"{fx} {nb}x{H}x{W}: {len(plan.steps)} steps, differing: {bad}")
