#!/usr/bin/env python3
"""Text-detector forward at the 1080p net input (960x544): op-by-op walk vs recorded launch-list replay, both programs.
One line per configuration; run under rocprofv3 --kernel-trace --stats for the GPU-side share."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_amd  # noqa: E402,F401
from vsr_amd.synth import make_det_weights as synthetic_weights  # noqa: E402
from vsr_amd.backend.tools import ocr_det  # noqa: E402
from vsr_amd.backend.tools.paddle_graph import load_graph  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1:] or ["ppocr_det_fast_graph.json", "ppocr_det_graph.json"]
for fx in which:
    g = load_graph(os.path.join(ROOT, "tests", "golden", fx))
    det = ocr_det.TextDetection(g, synthetic_weights(g), device=0)
    img = np.random.default_rng(3).integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    for tape in (False, True):
        det.use_tape = tape
        for _ in range(3):
            det.probability_map(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            det.probability_map(img)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        t1 = time.perf_counter()
        det.predict(img)
        pm = (time.perf_counter() - t1) * 1e3
        print(f"{fx}: forward at 1080p (960x544 net input), {'recorded launch list' if tape else 'op-by-op walk'}: {ms:.2f} ms/frame; "
              f"predict() incl. DB post-process on the host: {pm:.2f} ms/frame", flush=True)
    det.use_tape = True
    for nb in (4, 8, 16):
        imgs = [img] * nb
        for _ in range(3):
            det.probability_maps(imgs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            det.probability_maps(imgs)
        torch.cuda.synchronize()
        print(f"{fx}: {nb} frames per forward (recorded launch list): {(time.perf_counter() - t0) / 5 / nb * 1e3:.2f} ms/frame", flush=True)
