"""sttn-det at model resolution: HIP path vs oracle per frame for several batch lengths (diagnostic for the BASELINE-size test)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import vsr_amd  # noqa: F401
from oracle import cv2_restate as cv2r
from oracle.sttn_det import STTNDetOracle
from vsr_amd.engine import SttnEngine
from vsr_amd.synth import make_state_dict

torch.set_num_threads(32)
sd = make_state_dict(1, "det")
big = np.zeros((533, 1920, 1), np.uint8)
big[393:533, 278:1642] = 255
small = cv2r.resize_linear(big, (432, 240))[:, :, 0]
for L in [int(a) for a in sys.argv[1:]] or [12, 21]:
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, size=(L, 240, 432, 3), dtype=np.uint8)
    masks = np.stack([small] * L)
    eng = SttnEngine(sd, "det", device=0)
    comp, counts = eng.det_inpaint(torch.from_numpy(frames).cuda(), torch.from_numpy(masks).cuda())
    torch.cuda.synchronize()
    comp = comp.cpu().numpy()
    t = time.time()
    ref = np.stack([r.astype(np.float32) for r in STTNDetOracle(sd).inpaint(list(frames), list(masks))])
    d = np.abs(comp - ref)
    print(f"L={L}: oracle {time.time() - t:.0f} s; max|d| {d.max()}, differing {(d > 0).mean():.3e}", flush=True)
    for i in range(L):
        if d[i].max() > 1:
            print(f"   frame {i}: visits {counts[i]}, max|d| {d[i].max()}, differing {(d[i] > 0).mean():.3e}, >2: {(d[i] > 2).mean():.3e}")
    eng.close()
