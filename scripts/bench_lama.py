#!/usr/bin/env python3
"""Throughput of the LaMa engine (row a12) on one GPU: big-LaMa (18 FFC blocks) on 1080p strips (1920x360), mini-batches of 4 as
LamaInpaint._inpaint_batch hands them over (lama_inpaint.py:37), frames resident in HBM.  One JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_amd  # noqa: E402,F401
from vsr_amd.engine import LamaEngine  # noqa: E402
from vsr_amd.synth import make_lama_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--height", type=int, default=360)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--precision", default="f32", choices=["f32", "split"])
args = ap.parse_args()

eng = LamaEngine(make_lama_state_dict(0, 18), device=0)
eng.set_precision(args.precision)
B, H, W = args.batch, args.height, args.width
g = torch.Generator(device="cuda").manual_seed(3)
img = torch.randint(0, 256, (B, H, W, 3), device="cuda", generator=g, dtype=torch.uint8)
mask = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
mask[H // 2: H - 8, W // 8: W - W // 8] = 255
out = torch.empty_like(img)
for _ in range(args.warmup):
    eng.inpaint(img, mask, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    eng.inpaint(img, mask, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
fl = eng.flops(B, H, W)
print(json.dumps({"metric": f"LaMa inpainted strips/s ({W}x{H}, mini-batch {B})", "value": round(B / dt, 3), "unit": "frames/s",
                  "ms_per_call": round(dt * 1e3, 2), "tflops": round(fl / dt / 1e12, 2), "tflop_per_frame": round(fl / B / 1e12, 3),
                  "dtype": "f32" if args.precision == "f32" else "f32 (operands as fp16 hi/lo pairs)", "fallbacks": eng.fallbacks()}))
eng.close()
