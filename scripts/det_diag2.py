"""sttn-det BASELINE-size diagnosis: where do the HIP path and the oracle part ways on the 47-frame 1080p batch?"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import vsr_amd  # noqa: F401
from oracle import cv2_restate as cv2r
from oracle.sttn_det import STTNDetOracle
from tests import _baseline_oracle as bo
from vsr_amd.engine import SttnEngine
from vsr_amd.synth import make_state_dict

torch.set_num_threads(32)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 47
sd = make_state_dict(1, "det")
clip, mask, j = bo.job_inputs("det_1080p")
clip = clip[:L]
y0, y1, _, _ = bo.strip_rows("det_1080p")
o = STTNDetOracle(sd)
t = time.time()
fs = [cv2r.resize_linear(f[y0:y1], (432, 240)) for f in clip]
ms = [cv2r.resize_linear(mask[y0:y1, :, None], (432, 240))[:, :, 0] for _ in clip]
comps = o.inpaint(fs, ms)
print(f"oracle inpaint {time.time() - t:.0f} s", flush=True)
ref_comp = np.stack([c.astype(np.float32) for c in comps])
eng = SttnEngine(sd, "det", device=0)
# (b) network + model-resolution blend on the oracle's resized inputs
comp, counts = eng.det_inpaint(torch.from_numpy(np.stack(fs)).cuda(), torch.from_numpy(np.stack(ms)).cuda())
torch.cuda.synchronize()
d = np.abs(comp.cpu().numpy() - ref_comp)
print(f"(b) det_inpaint on the oracle's resized frames: max|d| {d.max()}, differing {(d > 0).mean():.3e}; counts {counts.tolist()}")
for i in range(L):
    if d[i].max() > 1:
        print(f"    frame {i}: visits {counts[i]} max {d[i].max()} differing {(d[i] > 0).mean():.3e}")
# (c) the whole batch call
areas = [(y0, y1, 0, clip.shape[2])]
dfr = torch.from_numpy(clip).cuda()
eng.det_batch(dfr, torch.from_numpy(mask).cuda(), areas)
torch.cuda.synchronize()
got = dfr.cpu().numpy()[:, y0:y1]
ref = np.stack([cv2r.resize_linear(comps[i], (clip.shape[2], y1 - y0)).astype(np.uint8)[:, :, ::-1] for i in range(L)])
d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
print(f"(c) det_batch vs oracle upscaled: max|d| {d.max()}, differing {(d > 0).mean():.3e}")
for i in range(L):
    print(f"    frame {i}: visits {counts[i]} dtype {comps[i].dtype} max {d[i].max()} differing {(d[i] > 0).mean():.3e} >2: {(d[i] > 2).mean():.2e}")
# (d) upscale alone: GPU upscale of the ORACLE's comp is not reachable through the ABI; compare instead the oracle's upscale of the GPU comp
gpu_comp = comp.cpu().numpy()
ref2 = np.stack([cv2r.resize_linear(gpu_comp[i].astype(np.uint8) if counts[i] == 1 else gpu_comp[i], (clip.shape[2], y1 - y0)).astype(np.uint8)[:, :, ::-1] for i in range(L)])
d2 = np.abs(got.astype(np.int16) - ref2.astype(np.int16))
print(f"(d) det_batch vs the oracle's upscale of the GPU comp: max|d| {d2.max()}, differing {(d2 > 0).mean():.3e}")
bad = np.argwhere(d2 > 2)
print("    first large ones (frame, y, x, c):", bad[:8].tolist())
eng.close()
