#!/usr/bin/env python3
"""Throughput of --inpaint-mode propainter on one GPU: PropainterInpaint.inpaint (RAFT 20 iterations -> flow completion ->
image propagation -> generator over sliding windows) on a clip of 1080p-strip crops (1920x360).  One JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_amd  # noqa: E402,F401
from vsr_amd.backend.inpaint.propainter_inpaint import PropainterInpaint  # noqa: E402
from vsr_amd.synth import make_clip, make_propainter_state_dict, make_raft_state_dict, make_rfc_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=30)
ap.add_argument("--height", type=int, default=360)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--precision", default="f32", choices=["f32", "split", "f16", "f16-raft-split"])
args = ap.parse_args()

H, W, n = args.height, args.width, args.frames
box = (H // 2, H - H // 6, W // 6, W - W // 6)
frames = list(make_clip(n, H, W, box, seed=4))
mask = np.zeros((H, W), np.uint8)
mask[box[0]:box[1], box[2]:box[3]] = 255
plug = PropainterInpaint("cuda:0", {"raft": make_raft_state_dict(0), "rfc": make_rfc_state_dict(0), "propainter": make_propainter_state_dict(0)}, precision=args.precision)
for _ in range(args.warmup):
    plug.inpaint(frames, mask)
torch.cuda.synchronize()
stages = {}
t0 = time.perf_counter()
for _ in range(args.steps):
    out = plug.inpaint(frames, mask)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
extra = {}
if args.precision != "f32":
    extra["fp32_fallback_calls"] = sum(e.fallbacks() for e in (plug.fix_raft, plug.fix_flow_complete, plug.model))
    for e in (plug.fix_raft, plug.fix_flow_complete, plug.model):
        e.set_precision("f32")
    ref = plug.inpaint(frames, mask)
    mse = float(np.mean((np.stack(out).astype(np.float64) - np.stack(ref).astype(np.float64)) ** 2))
    extra["psnr_db_vs_exact_mode"] = "inf" if mse == 0 else round(20 * np.log10(255.0 / np.sqrt(mse)), 2)
print(json.dumps({"metric": "propainter frames/s (1920x360 strip crops, host arrays in / out)", "value": round(n / dt, 3), "unit": "frames/s",
                  "frames": n, "s_per_call": round(dt, 3),
                  "dtype": {"f32": "f32", "split": "f32 (operands as fp16 hi/lo pairs)", "f16": "f16 operands, f32 accumulate (RAFT exact f32)", "f16-raft-split": "f16 operands, f32 accumulate (RAFT on fp16 hi/lo pairs)"}[args.precision], **extra, "raft_iters": plug.raft_iter,
                  "note": "one PropainterInpaint.inpaint call; includes H2D / D2H of the crops and the host-side u8 blending"}))
plug.close()
