"""One-off CPU fuzz of the ProPainter generator's box promise + per-frame encoder cache: random frame sizes, window
lengths and boxes; the cached / boxed plan's replay must equal the full plan's inside the box.  python scripts/r04/fuzz_pp_box_cache.py [n]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vsr_amd  # noqa: E402,F401
import _replay_pp as rp  # noqa: E402
from oracle.make_golden import propainter_inputs  # noqa: E402
from vsr_amd import _lib  # noqa: E402
from vsr_amd.engine import PpEngine  # noqa: E402
from vsr_amd.synth import make_propainter_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(7)
e = PpEngine(device=-1, state_dict=make_propainter_state_dict(0))
wts = e.packed_weights()
worst = 0.0
for it in range(n):
    H, W = int(rng.integers(5, 17)) * 8, int(rng.integers(5, 25)) * 8
    t = int(rng.integers(2, 7)); lt = int(rng.integers(2, t + 1))
    frames, masks, ff, fb = propainter_inputs(100 + it, t, lt, H, W)
    m8 = masks[:, 0].astype(np.uint8)
    sel = (frames * (1 - masks)).astype(np.float32)
    flags = e.window_flags(m8[:lt])
    y0 = int(rng.integers(0, H // 8)) * 8; y1 = int(rng.integers(y0 // 8 + 1, H // 8 + 1)) * 8
    x0 = int(rng.integers(0, W // 8)) * 8; x1 = int(rng.integers(x0 // 8 + 1, W // 8 + 1)) * 8
    full = rp.gen_plan_view(_lib, e, t, lt, H, W, flags)
    want, _ = rp.replay_gen(full, wts, sel, ff, fb, m8, m8, lt)
    full.close()
    ev = rp.gen_plan_view(_lib, e, t, t - lt, H, W, None, mode=1)          # the reference frames first (they need tokens), then the local ones
    order = list(range(lt, t)) + list(range(lt))
    feats_o, toks_o = rp.replay_encode(ev, wts, sel[order], m8[order], m8[order])
    ev.close()
    feats = np.zeros_like(feats_o); feats[order] = feats_o
    toks = np.zeros((t,) + toks_o.shape[1:], np.float32)
    if t > lt:
        toks[order[:t - lt]] = toks_o[:t - lt]
    cv = rp.gen_plan_view(_lib, e, t, lt, H, W, flags, box=(y0, y1, x0, x1), mode=2)
    got, _ = rp.replay_gen(cv, wts, sel, ff, fb, m8, m8, lt, cached=(feats, toks))
    saved = 1 - cv.flops / max(1.0, float(_lib.lib.vsr_pp_flops(e.handle, t, lt, H, W, np.ascontiguousarray(flags).ctypes.data, flags.size)))
    cv.close()
    d = float(np.abs(got[:, :, y0:y1, x0:x1] - want[:, :, y0:y1, x0:x1]).max())
    worst = max(worst, d)
    print(f"{it}: {H}x{W} t={t} lt={lt} box=({y0},{y1},{x0},{x1}) max|d|={d:.2e} window FLOPs -{100 * saved:.0f}%", flush=True)
    assert d <= 5e-5, "box / cache replay differs from the full plan"
e.close()
print("worst", worst)
