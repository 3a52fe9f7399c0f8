#!/usr/bin/env python3
"""One 68-frame propainter batch (exact fp32) with HIP events around every launch: ms per op tag (engine : kernel : tag), to compare
VSR_FLOW_THIN settings (flow_engine.hip thin_variant).  Prints the timed batch's wall time, a digest of the frames and the 40 largest keys."""
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vsr_amd  # noqa: E402,F401
from vsr_amd import engine as E  # noqa: E402
from vsr_amd.backend.inpaint.propainter_inpaint import PropainterInpaint  # noqa: E402
from vsr_amd.synth import make_clip, make_propainter_state_dict, make_raft_state_dict, make_rfc_state_dict  # noqa: E402

L, H, W = int(os.environ.get("AB_FRAMES", "68")), 360, 1920
box = (H // 2, H - H // 6, W // 6, W - W // 6)
base = make_clip(10, H, W, box, seed=4)
d = torch.from_numpy(base).cuda()
frames = torch.cat([torch.roll(d, shifts=(2 * k, 3 * k), dims=(1, 2)) for k in range((L + 9) // 10)], 0)[:L].contiguous()
mask = np.zeros((H, W), np.uint8)
mask[box[0]:box[1], box[2]:box[3]] = 255
plug = PropainterInpaint("cuda:0", {"raft": make_raft_state_dict(0), "rfc": make_rfc_state_dict(0), "propainter": make_propainter_state_dict(0)}, precision="f32")
out = plug.inpaint(frames, mask)
torch.cuda.synchronize()
best = 1e9
for _ in range(2):
    t0 = time.perf_counter()
    out = plug.inpaint(frames, mask)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"VSR_FLOW_THIN={os.environ.get('VSR_FLOW_THIN', '(default)')} tiles<={os.environ.get('VSR_FLOW_THIN_TILES', '768')} K<={os.environ.get('VSR_FLOW_THIN_K', '2600')}: "
      f"{best:.3f} s per {L}-frame batch = {L / best:.2f} fps; frames digest {digest}", flush=True)
plug.profile = {}
plug.inpaint(frames, mask)
plug.profile = {}
E.flow_timing_reset()
E.flow_timing(True)
plug.inpaint(frames, mask)
torch.cuda.synchronize()
E.flow_timing(False)
rows = []
for k in E.flow_timing_keys():
    ms, n, fl = E.flow_timing_get(k)
    rows.append((ms, n, fl, k))
rows.sort(reverse=True)
by_eng = {}
for ms, n, fl, k in rows:
    by_eng[k.split(":")[0]] = by_eng.get(k.split(":")[0], 0.0) + ms
print("  per engine (profiled call, one stream):", {k: round(v, 1) for k, v in by_eng.items()})
for ms, n, fl, k in rows[:int(os.environ.get("AB_ROWS", "45"))]:
    print(f"  {ms:9.2f} ms {n:6d} launches {fl / ms / 1e9 if ms > 0 else 0:7.1f} TF  {k}")
plug.close()
