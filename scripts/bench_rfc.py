#!/usr/bin/env python3
"""Throughput of the flow-completion stage (row a15) on one GPU: forward_bidirect_flow + combine_flow over the flows of a
clip of 1080p-strip frames (1920x360), inputs resident in HBM.  One JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_amd  # noqa: E402,F401
from vsr_amd.engine import RfcEngine  # noqa: E402
from vsr_amd.synth import make_rfc_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=21)
ap.add_argument("--height", type=int, default=360)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
args = ap.parse_args()

eng = RfcEngine(make_rfc_state_dict(0), device=0)
t, H, W = args.frames, args.height, args.width
g = torch.Generator(device="cuda").manual_seed(3)
ff = torch.randn((t - 1, 2, H, W), device="cuda", generator=g) * 3
fb = torch.randn((t - 1, 2, H, W), device="cuda", generator=g) * 3
masks = torch.zeros((t, H, W), dtype=torch.uint8, device="cuda")
masks[:, H // 2: H // 2 + H // 4, W // 8: W - W // 8] = 1
for _ in range(args.warmup):
    eng.complete(ff, fb, masks)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    eng.complete(ff, fb, masks)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
fl = eng.flops(t, H, W)
print(json.dumps({"metric": "flow completion, flow fields/s (1920x360, both directions)", "value": round(2 * (t - 1) / dt, 3),
                  "unit": "flow fields/s", "ms_per_call": round(dt * 1e3, 2), "frames": t, "tflops": round(fl / dt / 1e12, 2),
                  "gflop_per_flow_field": round(fl / (2 * (t - 1)) / 1e9, 1), "dtype": "f32"}))
eng.close()
