#!/usr/bin/env python3
"""DBPostProcess on the device, per map, on the map of the file-to-file runs (one full-width subtitle box at the 960x544 net input) and on
maps of several small boxes: wall time per map (one read-back each) and per batch of 8 (one read-back); run under rocprofv3 for the kernels."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vsr_amd  # noqa: E402,F401
from vsr_amd.backend.tools import ocr_det  # noqa: E402
from test_db_postprocess import blob_map  # noqa: E402

dev = torch.device("cuda:0")
H, W, rh, rw = 1080, 1920, 544, 960
ymin, ymax, xmin, xmax = 950, 1070, 288, 1632
sy, sx = rh / H, rw / W
hh, ww = (ymax - ymin) * sy, (xmax - xmin) * sx
inset = hh * ww * (1 - 0.16) / (2 * (hh + ww))
wide = torch.full((rh, rw), 0.02, dtype=torch.float32, device=dev)
wide[int(ymin * sy + inset):int(ymax * sy - inset) + 1, int(xmin * sx + inset):int(xmax * sx - inset) + 1] = 0.93
maps = {"one full-width box": wide, "6 boxes": torch.from_numpy(blob_map(1, rh, rw, 6, 0, 0, max_tilt=0.3)).to(dev),
        "60 boxes + specks": torch.from_numpy(blob_map(2, rh, rw, 60, 0, 12, max_tilt=0.3)).to(dev), "empty": torch.full((rh, rw), 0.02, device=dev),
        "all foreground (one 544-row component: host path)": torch.full((rh, rw), 0.9, device=dev),
        "noise, 50 % foreground (host path)": (torch.rand((rh, rw), device=dev) < 0.5).float() * 0.8 + 0.1,
        "noise, 10 % foreground (host path)": (torch.rand((rh, rw), device=dev) < 0.1).float() * 0.8 + 0.1}
post = ocr_det.DeviceDBPostProcess(dev)
for name, m in maps.items():
    for _ in range(3):
        b, s = post(m, H, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50 if "host path" not in name else 3
    for _ in range(reps):
        post(m, H, W)
    one = (time.perf_counter() - t0) / reps * 1e3
    if "host path" in name:                      # the device part alone (labelling + records), without the host fallback behind it
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            post._launch(m.contiguous(), H, W, 0.3, 0.6, 1.5, 3)
        torch.cuda.synchronize()
        print(f"{name}: device part {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per map", flush=True)
    if "host path" in name:
        print(f"{name}: {len(s)} boxes, {one:.1f} ms per map through the host fallback; fallbacks {post.host_fallbacks}", flush=True)
        continue
    m8 = torch.stack([m] * 8)
    for _ in range(2):
        post.batch(m8, H, W)
    t0 = time.perf_counter()
    for _ in range(10):
        post.batch(m8, H, W)
    eight = (time.perf_counter() - t0) / 80 * 1e3
    print(f"{name}: {len(s)} boxes, {one:.3f} ms per map (one read-back each), {eight:.3f} ms per map in batches of 8 (one read-back); fallbacks {post.host_fallbacks}", flush=True)
