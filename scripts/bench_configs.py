#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on ONE GPU (chunks resident in HBM, same timing rules as
bench.py, which keeps the headline config).  One JSON line per configuration; fills BASELINE.md section 4.

  config 2: 720p  sttn-auto, 50-frame chunks, fp32
  config 3: 1080p sttn-det, batch_generator sizes of a 1200-frame interval (25 x 47 + 25), fp32
  config 5: 4K    sttn-auto, 50-frame chunks, fp16 operands (per GPU; the 8-GPU run is the driver's)
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_amd  # noqa: E402,F401
from bench import RES, make_chunk_on_device  # noqa: E402
from vsr_amd.backend.tools.inpaint_tools import batch_generator, create_mask, get_inpaint_area_by_mask, threshold_mask  # noqa: E402
from vsr_amd.engine import SttnEngine  # noqa: E402
from vsr_amd.synth import make_state_dict  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def run_auto(name, res, precision, steps=4, warmup=1, L=50):
    H, W, box = RES[res]
    eng = SttnEngine(make_state_dict(0, "auto"), "auto", device=0, precision=precision)
    mask01 = threshold_mask(create_mask((H, W), [(box[2], box[3], box[0], box[1])]))
    areas = get_inpaint_area_by_mask(W, H, int(W * 3 / 16), mask01)
    dmask = torch.from_numpy(np.ascontiguousarray(mask01[:, :, 0])).cuda()
    src = make_chunk_on_device(L, H, W, box, seed=2, device=torch.device("cuda", 0))
    work = src.clone()

    def step():
        work.copy_(src)
        eng.auto_chunk(work, dmask, areas)

    dt = timed(step, steps, warmup)
    fps = steps * L / dt
    out = {"config": name, "mode": "sttn-auto", "res": res, "precision": precision, "chunk_frames": L, "fps": round(fps, 2),
           "ms_per_chunk": round(dt / steps * 1e3, 2), "model_tflops": round(eng.chunk_flops(L, dmask, areas) / L * fps / 1e12, 2),
           "fp32_fallback_chunks": eng.fallbacks()}
    eng.close()
    print(json.dumps(out), flush=True)


def run_det(name, res, precision, total=1200, reps=1):
    H, W, box = RES[res]
    eng = SttnEngine(make_state_dict(0, "det"), "det", device=0, precision=precision)
    mask = create_mask((H, W), [(box[2], box[3], box[0], box[1])])
    areas = get_inpaint_area_by_mask(W, H, int(W * 5 / 18), mask[:, :, None] if mask.ndim == 2 else mask)
    dmask = torch.from_numpy(np.ascontiguousarray(mask if mask.ndim == 2 else mask[:, :, 0])).cuda()
    sizes = [len(b) for b in batch_generator(list(range(total)), 50)]          # 25 x 47 + 25 for 1200 frames
    Lmax = max(sizes)
    src = make_chunk_on_device(Lmax, H, W, box, seed=3, device=torch.device("cuda", 0))
    work = src.clone()
    sample = sorted(set(sizes))                                               # time each distinct batch size, weight by count

    def make(L):
        def step():
            work[:L].copy_(src[:L])
            eng.det_batch(work[:L], dmask, areas)
        return step

    per = {}
    for L in sample:
        dt = timed(make(L), 3 if L == Lmax else 2, 1)
        per[L] = dt / (3 if L == Lmax else 2)
    wall = sum(per[L] for L in sizes)
    fps = total / wall
    flops = sum(eng.chunk_flops(L, dmask, areas) for L in sizes)          # what is contracted (last block / decoder rows trimmed to what is read)
    out = {"config": name, "mode": "sttn-det", "res": res, "precision": precision, "batches": f"{sizes.count(Lmax)}x{Lmax}+{sizes[-1]}",
           "fps": round(fps, 2), "ms_per_batch": {str(L): round(per[L] * 1e3, 2) for L in sample},
           "model_tflops": round(flops / wall / 1e12, 2), "gflop_per_frame": round(flops / total / 1e9, 1)}
    eng.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["2", "3", "5"]
    if "2" in which:
        run_auto("2: 720p sttn-auto fp32", "720p", "f32")
    if "3" in which:
        run_det("3: 1080p sttn-det fp32 (known box injected on every frame)", "1080p", "f32")
    if "5" in which:
        run_auto("5: 4K sttn-auto fp16 operands (one GPU of the 8)", "4k", "f16")
        run_auto("5: 4K sttn-auto fp32 (same, exact mode)", "4k", "f32")
