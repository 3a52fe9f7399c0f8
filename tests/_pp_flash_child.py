"""Child process of tests/test_gpu_pp.py::test_fused_window_attention_equals_three_ops: one generator forward on the GPU with this
process's VSR_PP_FLASH (the library reads it once); saves the tanh output."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vsr_amd  # noqa: E402,F401
from oracle.make_golden import propainter_inputs  # noqa: E402  (seeded inputs only; nothing is computed by the oracle here)
from vsr_amd.engine import PpEngine  # noqa: E402
from vsr_amd.synth import make_propainter_state_dict  # noqa: E402

t, lt, H, W, precision, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
frames, masks, ff, fb = propainter_inputs(80 + t, t, lt, H, W)
e = PpEngine(device=0, state_dict=make_propainter_state_dict(0))
if precision != "f32":
    e.set_precision(precision)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
m = masks[:, 0].astype(np.uint8)
y = e.forward(d(frames * (1 - masks)), d(ff), d(fb), d(m), d(m), lt)
torch.cuda.synchronize()
np.save(out, y.cpu().numpy())
print("FALLBACKS", e.fallbacks())
e.close()
