"""Child process of tests/test_gpu_sttn.py::test_shared_first_block_qkv_gives_the_same_frames: one sttn-auto chunk on the GPU with the
tuning of this process's environment (the library reads VSR_QKV0_SHARED once); prints the SHA-256 of the written frames."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vsr_amd  # noqa: E402,F401
from vsr_amd import synth  # noqa: E402
from vsr_amd.backend.tools.inpaint_tools import create_mask, get_inpaint_area_by_mask, threshold_mask  # noqa: E402
from vsr_amd.engine import SttnEngine  # noqa: E402

L, H, W, lanes, precision = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
box = (H * 5 // 6, H - 10, W // 8, W * 7 // 8)
eng = SttnEngine(synth.make_state_dict(1, "auto"), "auto", device=0)
eng.set_lanes(lanes)
if precision:
    eng.set_precision({1: "split", 2: "split-format", 3: "f16"}[precision])
frames = torch.from_numpy(synth.make_clip(L, H, W, box, seed=21)).cuda()
m01 = threshold_mask(create_mask((H, W), [(box[2], box[3], box[0], box[1])]))
areas = get_inpaint_area_by_mask(W, H, int(W * 3 / 16), m01)
dmask = torch.from_numpy(np.ascontiguousarray(m01[:, :, 0])).cuda()
before = frames.clone()
eng.auto_chunk(frames, dmask, areas)
torch.cuda.synchronize()
assert not torch.equal(frames, before)
print("DIGEST", hashlib.sha256(frames.cpu().numpy().tobytes()).hexdigest(), eng.chunk_flops(L, dmask, areas))
