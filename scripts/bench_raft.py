#!/usr/bin/env python3
"""Throughput of the RAFT stage (row a14) on one GPU: RAFT_bi over a clip of 1080p-strip frames (1920x360, the size
--inpaint-mode propainter hands to RAFT), 20 iterations, frames resident in HBM.  One JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_amd  # noqa: E402,F401
from vsr_amd.engine import RaftEngine  # noqa: E402
from vsr_amd.synth import make_flow_frames, make_raft_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=11)
ap.add_argument("--height", type=int, default=360)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
args = ap.parse_args()

eng = RaftEngine(make_raft_state_dict(0), device=0)
d = torch.from_numpy(make_flow_frames(args.frames, args.height, args.width, seed=9)).cuda()
for _ in range(args.warmup):
    eng.flows(d, iters=args.iters)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    eng.flows(d, iters=args.iters)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
pairs = 2 * (args.frames - 1)
fl = eng.flops(args.frames, args.height, args.width, args.iters)
print(json.dumps({"metric": "RAFT pair-directions/s (1920x360, 20 iterations)", "value": round(pairs / dt, 3), "unit": "pair-directions/s",
                  "ms_per_call": round(dt * 1e3, 2), "frames": args.frames, "pair_directions": pairs, "iters": args.iters,
                  "tflops": round(fl / dt / 1e12, 2), "tflop_per_pair_direction": round(fl / pairs / 1e12, 3), "dtype": "f32"}))
eng.close()
