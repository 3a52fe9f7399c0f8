"""The MI355X plugins against what the REFERENCE's own wrapper code produced (tests/golden/wrappers.npz, written by
oracle/make_golden_wrappers.py executing backend/inpaint/sttn_auto_inpaint.py and sttn_det_inpaint.py from /root/reference).
No oracle restatement sits in between here: plugin output vs reference output on the same frames, mask and weights.
Covers SURVEY row a2 (STTNInpaint.__call__) next to a1 / a3 / a10.  Bar: PSNR >= 50 dB on the repainted strip, max |d| <= 2
grey levels (u8 truncation flips of an fp32 sum in another order), everything outside bit-identical.
"""
import json
import os

import numpy as np
import pytest

from oracle.make_golden_wrappers import AUTO_AB, AUTO_CLIP, DET_CLIP, STTN_CFG
from oracle.sttn_auto import calculate_psnr
from vsr_amd.backend.config import config
from vsr_amd.backend.tools.inpaint_tools import create_mask
from vsr_amd.synth import make_clip, make_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    with np.load(os.path.join(GOLD, "wrappers.npz")) as z:
        arrays = {k: z[k] for k in z.files}
    return arrays, json.load(open(os.path.join(GOLD, "wrappers.json")))


@pytest.fixture()
def sttn_cfg():
    keys = ("sttnNeighborStride", "sttnReferenceLength", "sttnMaxLoadNum")
    old = {k: getattr(config, k).value for k in keys}
    for k in keys:
        getattr(config, k).value = STTN_CFG[k]
    yield
    for k in keys:
        getattr(config, k).value = old[k]


def _bar(got, ref, what):
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    psnr = calculate_psnr(got, ref)
    print(f"{what}: PSNR vs the reference's output {psnr:.2f} dB, max|d| {d.max()}, differing {float((d > 0).mean()):.2e}")
    assert got.shape == ref.shape and psnr >= 50.0 and d.max() <= 2


def _auto_inputs():
    c = AUTO_CLIP
    clip = make_clip(c["n"], c["H"], c["W"], c["box"], seed=c["seed"])
    b = c["box"]
    return clip, create_mask((c["H"], c["W"]), [(b[2], b[3], b[0], b[1])])


def test_sttn_inpaint_vs_reference(built_lib, gpu_device, gold, sttn_cfg):
    """STTNInpaint.inpaint (a3): which frames stay uint8, and the composites themselves."""
    from oracle import cv2_restate as cv2r
    from vsr_amd.backend.inpaint.sttn_auto_inpaint import STTNInpaint

    z, js = gold
    clip, _ = _auto_inputs()
    y0, y1 = js["auto_area"]
    plug = STTNInpaint("cuda:0", {"netG": make_state_dict(0, "auto")})
    comps = plug.inpaint([cv2r.resize_linear(f[y0:y1], (640, 120)) for f in clip[:6]])
    plug.engine.close()
    assert [str(x.dtype) for x in comps] == js["auto_inpaint_dtypes"]
    _bar(np.stack([comps[i].astype(np.float32) for i in (0, 2, 5)]), z["auto_inpaint_x4"].astype(np.float32) / 4, "STTNInpaint.inpaint")


def test_sttn_plugin_call_vs_reference(built_lib, gpu_device, gold, sttn_cfg):
    """STTNInpaint.__call__(frames, mask) -> frames (a2): list in, list out, inputs untouched."""
    from vsr_amd.backend.inpaint.sttn_auto_inpaint import STTNInpaint

    z, js = gold
    clip, mask = _auto_inputs()
    plug = STTNInpaint("cuda:0", {"netG": make_state_dict(0, "auto")})
    frames_in = [f.copy() for f in clip[:6]]
    out = plug(frames_in, mask)
    plug.engine.close()
    assert isinstance(out, list) and len(out) == 6 and all(o.dtype == np.uint8 and o.shape == clip[0].shape for o in out)
    assert all(np.array_equal(a, b) for a, b in zip(frames_in, clip[:6])), "inputs are not mutated"
    out = np.stack(out)
    y0, y1 = js["auto_area"]
    assert np.array_equal(out[:, :y0], clip[:6, :y0]) and np.array_equal(out[:, y1:], clip[:6, y1:])
    _bar(out[:, y0:y1], z["auto_plugin_strip"], "STTNInpaint.__call__")
    assert plug([], mask) == []


@pytest.mark.parametrize("tag", ["all", "ab"])
def test_sttn_auto_call_vs_reference(built_lib, gpu_device, gold, sttn_cfg, tag):
    """STTNAutoInpaint.__call__ (a1): two chunks (7 + 5 frames), frame selection by A/B sections, writer / progress hooks."""
    from vsr_amd.backend.inpaint.sttn_auto_inpaint import STTNAutoInpaint
    from vsr_amd.backend.tools.video_io import ArrayVideo, ArrayWriter

    z, js = gold
    clip, mask = _auto_inputs()

    class Host:
        gui_mode = False
        ab_sections = None if tag == "all" else [range(a, e) for a, e in AUTO_AB]
        video_writer = ArrayWriter()
        ticks = 0

        def update_progress(self, tbar, increment):
            Host.ticks += increment

    plug = STTNAutoInpaint("cuda:0", {"netG": make_state_dict(0, "auto")}, ArrayVideo(clip.copy()))
    assert plug.clip_gap == 7
    plug(input_mask=mask, input_sub_remover=Host(), tbar=object())
    plug.sttn_inpaint.engine.close()
    out = np.stack(Host.video_writer.frames)
    assert out.shape == clip.shape and Host.ticks == clip.shape[0]
    y0, y1 = js["auto_area"]
    assert np.array_equal(out[:, :y0], clip[:, :y0]) and np.array_equal(out[:, y1:], clip[:, y1:])
    if tag == "ab":
        untouched = [j for j in range(clip.shape[0]) if not any(j in range(a, e) for a, e in AUTO_AB)]
        assert np.array_equal(out[untouched], clip[untouched])
    _bar(out[:, y0:y1], z[f"auto_call_{tag}_strip"], f"STTNAutoInpaint.__call__ ({tag})")


def test_sttn_det_call_vs_reference(built_lib, gpu_device, gold, sttn_cfg):
    """STTNDetInpaint.__call__ (a10)."""
    from vsr_amd.backend.inpaint.sttn_det_inpaint import STTNDetInpaint

    z, js = gold
    c = DET_CLIP
    clip = make_clip(c["n"], c["H"], c["W"], c["box"], seed=c["seed"])
    b = c["box"]
    mask = create_mask((c["H"], c["W"]), [(b[2], b[3], b[0], b[1])])
    plug = STTNDetInpaint("cuda:0", {"netG": make_state_dict(1, "det")})
    frames_in = [f.copy() for f in clip]
    out = np.stack(plug(frames_in, mask))
    plug.engine.close()
    assert all(np.array_equal(a, b2) for a, b2 in zip(frames_in, clip))
    y0, y1 = js["det_area"]
    assert np.array_equal(out[:, :y0], clip[:, :y0]) and np.array_equal(out[:, y1:], clip[:, y1:])
    _bar(out[:, y0:y1], z["det_call_strip"], "STTNDetInpaint.__call__")


def test_propainter_call_vs_reference(built_lib, gpu_device):
    """PropainterInpaint.__call__ (a13) against the reference's own wrapper + modules run on the CPU (wrappers_propainter.npz):
    288x704 clip, strip aligned to x8, 20 RAFT iterations, u8 overlap blending."""
    from oracle.make_golden_wrappers import PP_CLIP
    from vsr_amd.backend.inpaint.propainter_inpaint import PropainterInpaint
    from vsr_amd.synth import make_propainter_state_dict, make_raft_state_dict, make_rfc_state_dict

    with np.load(os.path.join(GOLD, "wrappers_propainter.npz")) as z:
        bbox, want, changed = z["bbox"], z["out"], z["changed"]
    c = PP_CLIP
    clip = make_clip(c["n"], c["H"], c["W"], c["box"], seed=c["seed"])
    b = c["box"]
    mask = create_mask((c["H"], c["W"]), [(b[2], b[3], b[0], b[1])])
    plug = PropainterInpaint("cuda:0", {"raft": make_raft_state_dict(0), "rfc": make_rfc_state_dict(0), "propainter": make_propainter_state_dict(0)},
                             sub_video_length=70)
    frames_in = [f.copy() for f in clip]
    out = np.stack(plug(frames_in, mask))
    plug.close()
    assert all(np.array_equal(a, b2) for a, b2 in zip(frames_in, clip))
    ch = np.unpackbits(changed)[: c["H"] * c["W"]].reshape(c["H"], c["W"]).astype(bool)
    assert np.array_equal(out[:, ~ch], clip[:, ~ch]), "pixels the reference leaves alone stay bit-identical"
    _bar(out[:, bbox[0]:bbox[1], bbox[2]:bbox[3]], want, "PropainterInpaint.__call__")
