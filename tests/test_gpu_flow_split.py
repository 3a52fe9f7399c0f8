"""Split-half arithmetic of the ProPainter stage engines (vsr_{raft,rfc,pp}_set_precision(1)): fp16 hi/lo operand pairs with
fp32 accumulation on the f16 matrix cores, range-guarded.  Checked against the exact fp32 mode of the same engine on the same
device (the exact mode is the one pinned to the oracles by test_gpu_{raft,rfc,pp}.py); tolerances are those of the exact mode
against its oracle, i.e. the split mode stays inside the noise the parity tests already allow."""
import numpy as np
import pytest
import torch

from oracle.make_golden import propainter_inputs, rfc_inputs
from vsr_amd.engine import PpEngine, RaftEngine, RfcEngine
from vsr_amd.synth import make_flow_frames, make_propainter_state_dict, make_raft_state_dict, make_rfc_state_dict

pytestmark = pytest.mark.gpu


def test_raft_split_mode(built_lib, gpu_device):
    e = RaftEngine(make_raft_state_dict(0), device=0)
    try:
        d = torch.from_numpy(make_flow_frames(3, 128, 192, seed=5)).to(gpu_device)
        f0, b0 = (x.clone() for x in e.flows(d, iters=20))
        e.set_precision("split")
        f1, b1 = (x.clone() for x in e.flows(d, iters=20))
        torch.cuda.synchronize()
        err = max((f1 - f0).abs().max().item(), (b1 - b0).abs().max().item())
        print(f"RAFT split vs exact, 20 iterations: max abs diff {err:.3e} px (flow range {f0.abs().max().item():.1f} px), fallbacks {e.fallbacks()}")
        assert torch.isfinite(f1).all() and e.fallbacks() == 0
        assert err <= 5e-3 and not torch.equal(f0, f1)
        e.set_precision("f32")
        f2, _ = e.flows(d, iters=20)
        assert torch.equal(f2, f0), "back in exact mode the result is the exact one"
        with pytest.raises(ValueError):
            e.set_precision("bf16")
    finally:
        e.close()


def test_rfc_split_mode(built_lib, gpu_device):
    e = RfcEngine(make_rfc_state_dict(0), device=0)
    try:
        ff, fb, masks = rfc_inputs(17, 6, 72, 104)
        m8 = torch.from_numpy((masks[:, 0] > 0).astype(np.uint8)).to(gpu_device)
        a = [torch.from_numpy(ff).to(gpu_device), torch.from_numpy(fb).to(gpu_device), m8]
        of0, ob0 = (x.clone() for x in e.complete(*a))
        e.set_precision("split")
        of1, ob1 = (x.clone() for x in e.complete(*a))
        torch.cuda.synchronize()
        err = max((of1 - of0).abs().max().item(), (ob1 - ob0).abs().max().item())
        print(f"flow completion split vs exact: max abs diff {err:.3e} (range {of0.abs().max().item():.1f}), fallbacks {e.fallbacks()}")
        assert e.fallbacks() == 0 and err <= 1e-3
        hole = torch.from_numpy(np.broadcast_to(masks[:-1] > 0, ff.shape).copy()).to(gpu_device)
        assert torch.equal(of1[~hole], a[0][~hole])
    finally:
        e.close()


def test_rfc_range_guard_falls_back_to_fp32(built_lib, gpu_device):
    """flows of 1e6 px leave the fp16 range: the call is redone with exact contractions and returns the exact result"""
    e = RfcEngine(make_rfc_state_dict(0), device=0)
    try:
        ff, fb, masks = rfc_inputs(18, 3, 64, 64)
        ff, fb = ff * 1e5, fb * 1e5
        m8 = torch.from_numpy((masks[:, 0] > 0).astype(np.uint8)).to(gpu_device)
        a = [torch.from_numpy(ff).to(gpu_device), torch.from_numpy(fb).to(gpu_device), m8]
        of0, ob0 = (x.clone() for x in e.complete(*a))
        e.set_precision("split")
        of1, ob1 = (x.clone() for x in e.complete(*a))
        torch.cuda.synchronize()
        assert e.fallbacks() == 1
        assert torch.equal(of0, of1) and torch.equal(ob0, ob1)
    finally:
        e.close()


def test_generator_split_mode(built_lib, gpu_device):
    e = PpEngine(device=0, state_dict=make_propainter_state_dict(0))
    try:
        t, lt, H, W = 5, 3, 64, 96
        frames, masks, ff, fb = propainter_inputs(83, t, lt, H, W)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
        m8 = masks[:, 0].astype(np.uint8)
        args = (d(frames * (1 - masks)), d(ff), d(fb), d(m8), d(m8), lt)
        o0 = e.forward(*args).clone()
        e.set_precision("split")
        o1 = e.forward(*args).clone()
        torch.cuda.synchronize()
        err = (o1 - o0).abs().max().item()
        print(f"generator split vs exact: max abs diff {err:.3e} (tanh output), fallbacks {e.fallbacks()}")
        assert e.fallbacks() == 0 and torch.isfinite(o1).all()
        assert err <= 2e-3
    finally:
        e.close()


# ---- mode 2: fp16 operands (fp32 tensors rounded on their way into the matrix cores), fp32 accumulation: the arithmetic class of the
# reference's `.half()` flow-completion and generator modules (propainter_inpaint.py:140-146,249-251).  Held to the exact mode of the
# same engine; the end-to-end bar (>= 50 dB on the frames against the fp32 CPU oracle) is tests/test_gpu_zbaseline.py's.
def test_rfc_f16_mode(built_lib, gpu_device):
    e = RfcEngine(make_rfc_state_dict(0), device=0)
    try:
        ff, fb, masks = rfc_inputs(17, 6, 72, 104)
        m8 = torch.from_numpy((masks[:, 0] > 0).astype(np.uint8)).to(gpu_device)
        a = [torch.from_numpy(ff).to(gpu_device), torch.from_numpy(fb).to(gpu_device), m8]
        of0, ob0 = (x.clone() for x in e.complete(*a))
        e.set_precision("f16")
        of1, ob1 = (x.clone() for x in e.complete(*a))
        torch.cuda.synchronize()
        err = max((of1 - of0).abs().max().item(), (ob1 - ob0).abs().max().item())
        print(f"flow completion f16 vs exact: max abs diff {err:.3e} px (range {of0.abs().max().item():.1f}), fallbacks {e.fallbacks()}")
        assert e.fallbacks() == 0 and torch.isfinite(of1).all() and not torch.equal(of0, of1)
        assert err <= 0.05, "completed flows within a twentieth of a pixel of the exact mode"
        hole = torch.from_numpy(np.broadcast_to(masks[:-1] > 0, ff.shape).copy()).to(gpu_device)
        assert torch.equal(of1[~hole], a[0][~hole])
        e.set_precision("f32")
        of2, _ = e.complete(*a)
        assert torch.equal(of2, of0), "back in exact mode the result is the exact one"
    finally:
        e.close()


def test_rfc_f16_range_guard_falls_back_to_fp32(built_lib, gpu_device):
    e = RfcEngine(make_rfc_state_dict(0), device=0)
    try:
        ff, fb, masks = rfc_inputs(18, 3, 64, 64)
        ff, fb = ff * 1e5, fb * 1e5
        m8 = torch.from_numpy((masks[:, 0] > 0).astype(np.uint8)).to(gpu_device)
        a = [torch.from_numpy(ff).to(gpu_device), torch.from_numpy(fb).to(gpu_device), m8]
        of0, ob0 = (x.clone() for x in e.complete(*a))
        e.set_precision("f16")
        of1, ob1 = (x.clone() for x in e.complete(*a))
        torch.cuda.synchronize()
        assert e.fallbacks() == 1
        assert torch.equal(of0, of1) and torch.equal(ob0, ob1)
    finally:
        e.close()


def test_generator_f16_mode(built_lib, gpu_device):
    e = PpEngine(device=0, state_dict=make_propainter_state_dict(0))
    try:
        t, lt, H, W = 5, 3, 64, 96
        frames, masks, ff, fb = propainter_inputs(83, t, lt, H, W)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
        m8 = masks[:, 0].astype(np.uint8)
        args = (d(frames * (1 - masks)), d(ff), d(fb), d(m8), d(m8), lt)
        o0 = e.forward(*args).clone()
        e.set_precision("f16")
        o1 = e.forward(*args).clone()
        torch.cuda.synchronize()
        err = (o1 - o0).abs().max().item()
        rms = (o1 - o0).pow(2).mean().sqrt().item()
        psnr = 20 * np.log10(2.0 / max(rms, 1e-12))
        print(f"generator f16 vs exact: max abs diff {err:.3e}, rms {rms:.3e} (tanh output in [-1, 1]: {psnr:.1f} dB), fallbacks {e.fallbacks()}")
        assert e.fallbacks() == 0 and torch.isfinite(o1).all() and not torch.equal(o0, o1)
        assert psnr >= 50.0
    finally:
        e.close()
