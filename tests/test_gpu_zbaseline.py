"""Parity at BASELINE.json's sizes: the HIP path (through the C-ABI) against the CPU oracle on full-size units.

  * vsr_sttn_auto_chunk on one 50-frame chunk with the real schedule (stride 5, references every 10: windows of T = 10..15,
    split-K of the coarse attention scales, frames visited 1 / 2 / 3 times) at 720p, 1080p and -- in the fp16-operand mode of
    BASELINE config 5, against the fp32 oracle -- at 4K;
  * STTNDetInpaint.__call__ on a 47-frame 1080p batch (what batch_generator(1200, 50) hands the plugin), strip 1920x533;
  * PropainterInpaint.__call__ on a 20-frame 1080p batch (strip 1920x360), 20 RAFT iterations.
The oracle runs take 30 s .. 4 min each; tests/_baseline_oracle.py runs them in background processes started at collection
time (tests/conftest.py), so this file -- alphabetically the last GPU file -- mostly finds them finished.
Bar (BASELINE.json north_star): PSNR >= 50 dB against the reference arithmetic on the repainted pixels, every other pixel
bit-identical; for the exact-fp32 modes additionally max |d| <= 2 grey levels (u8 truncation flips only).
"""
import numpy as np
import pytest
import torch

from oracle import cv2_restate as cv2r
from oracle.sttn_auto import calculate_psnr, get_inpaint_area_by_mask
from tests import _baseline_oracle as bo
from vsr_amd.synth import make_state_dict

pytestmark = [pytest.mark.gpu, pytest.mark.baseline_oracle]

PSNR_MIN_DB = 50.0


@pytest.fixture(autouse=True)
def reference_defaults():
    """the plugins read the process-global config: pin the reference's defaults (backend/config.py: stride 5, references every
    10, 50 / 70 frames per call) whatever an earlier test left behind"""
    from vsr_amd.backend.config import config

    keys = {"sttnNeighborStride": 5, "sttnReferenceLength": 10, "sttnMaxLoadNum": 50, "propainterMaxLoadNum": 70}
    old = {k: getattr(config, k).value for k in keys}
    for k, v in keys.items():
        getattr(config, k).value = v
    yield
    for k, v in old.items():
        getattr(config, k).value = v


def _report(name, got, ref, m=None):
    a, b = (got, ref) if m is None else (got[:, m], ref[:, m])
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    psnr = calculate_psnr(a, b)
    print(f"{name}: PSNR {psnr:.2f} dB over {a.size} repainted values, max|d| {d.max()}, differing {float((d > 0).mean()):.2e}")
    return psnr, int(d.max())


@pytest.mark.parametrize("name,precision", [("auto_720p", "f32"), ("auto_1080p", "f32"), ("auto_4k", "f16"), ("auto_4k", "f32")])
def test_auto_chunk_L50_vs_oracle(built_lib, gpu_device, name, precision):
    from vsr_amd.engine import SttnEngine

    clip, mask, j = bo.job_inputs(name)
    H, W = j["H"], j["W"]
    mask01 = cv2r.threshold_binary(mask, 127, 1)
    areas = get_inpaint_area_by_mask(W, H, int(W * 3 / 16), mask01[:, :, None])
    y0, y1 = areas[0][:2]
    eng = SttnEngine(make_state_dict(0, "auto"), "auto", device=0, precision=precision)       # default schedule: stride 5, refs every 10
    d = torch.from_numpy(clip).to(gpu_device)
    eng.auto_chunk(d, torch.from_numpy(mask01).to(gpu_device), areas)
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    assert eng.fallbacks() == 0
    eng.close()
    ref = bo.result(name)
    m = mask01.astype(bool)
    assert np.array_equal(got[:, ~m], clip[:, ~m]), "pixels outside the mask must be untouched"
    assert (got[:, m] != clip[:, m]).mean() > 0.5
    psnr, dmax = _report(f"{name} [{precision}] L={j['L']}", got[:, y0:y1], ref, m[y0:y1])
    assert psnr >= PSNR_MIN_DB
    if precision == "f32":
        assert dmax <= 2


def test_det_batch_L47_vs_oracle(built_lib, gpu_device):
    from vsr_amd.backend.inpaint.sttn_det_inpaint import STTNDetInpaint

    clip, mask, j = bo.job_inputs("det_1080p")
    y0, y1, _, _ = bo.strip_rows("det_1080p")
    assert y1 - y0 == 533
    plug = STTNDetInpaint("cuda:0", {"netG": make_state_dict(1, "det")})
    got = np.stack(plug([f for f in clip], mask))
    plug.engine.close()
    ref = bo.result("det_1080p")
    assert np.array_equal(got[:, :y0], clip[:, :y0]) and np.array_equal(got[:, y1:], clip[:, y1:])
    psnr, dmax = _report("sttn-det 1080p L=47 (strip 1920x533)", got[:, y0:y1], ref)
    assert psnr >= PSNR_MIN_DB and dmax <= 2


def test_det_portrait_vs_oracle(built_lib, gpu_device):
    """the portrait branch of STTNDetInpaint.__call__ (sttn_det_inpaint.py:48-51): split_h = int(H*5/9) on a 1080x1920 clip"""
    from vsr_amd.backend.inpaint.sttn_det_inpaint import STTNDetInpaint

    clip, mask, j = bo.job_inputs("det_portrait")
    y0, y1, _, _ = bo.strip_rows("det_portrait")
    assert y1 - y0 == int(1920 * 5 / 9) == 1066
    plug = STTNDetInpaint("cuda:0", {"netG": make_state_dict(1, "det")})
    got = np.stack(plug([f for f in clip], mask))
    plug.engine.close()
    ref = bo.result("det_portrait")
    assert np.array_equal(got[:, :y0], clip[:, :y0]) and np.array_equal(got[:, y1:], clip[:, y1:])
    psnr, dmax = _report("sttn-det portrait 1080x1920 L=25 (strip 1080x1066)", got[:, y0:y1], ref)
    assert psnr >= PSNR_MIN_DB and dmax <= 2


@pytest.mark.parametrize("name", ["pp_1080p_L68", "pp_1080p_L44"])
@pytest.mark.parametrize("switches", ["default", "box+cache", "f16", "f16-raft-split"])
def test_propainter_config4_batches_vs_oracle(built_lib, gpu_device, monkeypatch, name, switches):
    """BASELINE configs[3] at the batch sizes batch_generator(1200, 70) produces (17 x 68 + 44 frames): the plugin against the CPU
    oracle's run, which is committed as a fixture (tests/_baseline_oracle.py --fixture: every frame's repainted box sub-sampled,
    two frames in full, and the set of pixels the oracle changed)."""
    from vsr_amd.backend.inpaint.propainter_inpaint import PropainterInpaint
    from vsr_amd.synth import make_propainter_state_dict, make_raft_state_dict, make_rfc_state_dict

    on = "1" if switches == "box+cache" else "0"
    monkeypatch.setenv("VSR_PP_DECODE_BOX", on)
    monkeypatch.setenv("VSR_PP_ENC_CACHE", on)
    import os

    if not os.path.exists(os.path.join(bo.GOLDEN, name + ".npz")):
        pytest.skip(f"tests/golden/{name}.npz not generated (python -m tests._baseline_oracle --fixture {name})")
    fx = bo.fixture(name)
    clip, mask, j = bo.job_inputs(name)
    y0, y1, x0, x1 = bo.strip_rows(name)
    assert (y1 - y0, x1 - x0) == (360, 1920)
    sds = {"raft": make_raft_state_dict(0), "rfc": make_rfc_state_dict(0), "propainter": make_propainter_state_dict(0)}
    # "f16": the reference's GPU arithmetic (RAFT exact fp32, flow completion + generator on fp16 operands, fp32 accumulation) against
    # the fp32 CPU oracle; "f16-raft-split" additionally RAFT on fp16 hi/lo pairs.  Same >= 50 dB bar.
    plug = PropainterInpaint("cuda:0", sds, precision=switches if switches.startswith("f16") else "f32")
    got = np.stack(plug([f for f in clip], mask))
    fb = [e.fallbacks() for e in (plug.fix_raft, plug.fix_flow_complete, plug.model)]
    plug.close()
    assert fb == [0, 0, 0], f"range-guard fallbacks {fb}"
    assert np.array_equal(got[:, :y0], clip[:, :y0]) and np.array_equal(got[:, y1:], clip[:, y1:])
    rows, src = got[:, y0:y1], clip[:, y0:y1]
    changed = fx["changed"]
    assert np.array_equal(rows[:, ~changed], src[:, ~changed]), "pixels the reference leaves alone stay bit-identical"
    by0, by1, bx0, bx1 = fx["box"]
    st = int(fx["stride"])
    m = changed[by0:by1:st, bx0:bx1:st]
    psnr, dmax = _report(f"propainter 1080p L={j['L']} [{switches}] every frame, stride-{st} sample", rows[:, by0:by1:st, bx0:bx1:st], fx["sample"], m)
    assert psnr >= PSNR_MIN_DB
    ff = fx["full_frames"]
    psnr, dmax = _report(f"propainter 1080p L={j['L']} [{switches}] frames {ff.tolist()} in full", rows[ff, by0:by1, bx0:bx1], fx["full"],
                         changed[by0:by1, bx0:bx1])
    assert psnr >= PSNR_MIN_DB


def test_propainter_batch_L20_vs_oracle(built_lib, gpu_device):
    from vsr_amd.backend.inpaint.propainter_inpaint import PropainterInpaint
    from vsr_amd.synth import make_propainter_state_dict, make_raft_state_dict, make_rfc_state_dict

    clip, mask, j = bo.job_inputs("pp_1080p")
    y0, y1, x0, x1 = bo.strip_rows("pp_1080p")
    assert (y1 - y0, x1 - x0) == (360, 1920)
    sds = {"raft": make_raft_state_dict(0), "rfc": make_rfc_state_dict(0), "propainter": make_propainter_state_dict(0)}
    plug = PropainterInpaint("cuda:0", sds)
    assert plug.raft_iter == 20
    got = np.stack(plug([f for f in clip], mask))
    plug.close()
    ref = bo.result("pp_1080p")
    assert np.array_equal(got[:, :y0], clip[:, :y0]) and np.array_equal(got[:, y1:], clip[:, y1:])
    changed = (ref != clip[:, y0:y1]).any(axis=(0, 3))
    assert changed.any()
    assert np.array_equal(got[:, y0:y1][:, ~changed], clip[:, y0:y1][:, ~changed]), "pixels the reference leaves alone stay bit-identical"
    psnr, dmax = _report("propainter 1080p L=20 (strip 1920x360, 20 RAFT iterations)", got[:, y0:y1], ref, changed)
    assert psnr >= PSNR_MIN_DB
