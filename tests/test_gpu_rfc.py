"""Recurrent flow completion on the MI355X: vsr_rfc_complete through the C-ABI against oracle/rfc.py
(pinned to the reference module by tests/test_oracle_golden.py; torchvision.ops.deform_conv2d restated)."""
import numpy as np
import pytest
import torch

from oracle.make_golden import rfc_inputs
from oracle.rfc import RfcOracle
from vsr_amd.engine import RfcEngine
from vsr_amd.synth import make_rfc_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rfc_sd():
    return make_rfc_state_dict(0)


@pytest.fixture(scope="module")
def engine(rfc_sd, built_lib, gpu_device):
    e = RfcEngine(rfc_sd, device=0)
    yield e
    e.close()


def _run(engine, gpu_device, ff, fb, masks):
    m8 = torch.from_numpy((masks[:, 0] > 0).astype(np.uint8)).to(gpu_device)
    of, ob = engine.complete(torch.from_numpy(ff).to(gpu_device), torch.from_numpy(fb).to(gpu_device), m8)
    torch.cuda.synchronize()
    return of.cpu().numpy(), ob.cpu().numpy()


@pytest.mark.parametrize("t,H,W", [(2, 64, 64), (6, 72, 104), (4, 360, 640)])
def test_completion_matches_oracle(engine, rfc_sd, gpu_device, t, H, W):
    ff, fb, masks = rfc_inputs(11 + t, t, H, W)
    of, ob = _run(engine, gpu_device, ff, fb, masks)
    cf, cb, _, _ = RfcOracle(rfc_sd).complete_bi(torch.from_numpy(ff), torch.from_numpy(fb), torch.from_numpy(masks))
    hole = np.broadcast_to(masks[:-1] > 0, of.shape)
    for name, got, ref in (("forward", of, cf.numpy()), ("backward", ob, cb.numpy())):
        err = np.abs(got - ref).max()
        print(f"{name} {t}x{H}x{W}: max abs err {err:.3e}, range {np.abs(ref).max():.1f}")
        assert np.isfinite(got).all()
        assert err <= 1e-3, f"{name}: max abs err {err:.3e}"
    assert np.array_equal(of[~hole], ff[~hole]), "outside the hole the flow passes through bit for bit"


def test_strip_size_properties(engine, gpu_device):
    """1080p strip (1920x360): deterministic, pass-through outside the hole, and the workspace survives plan changes."""
    ff, fb, masks = rfc_inputs(31, 3, 360, 1920)
    small = rfc_inputs(32, 2, 64, 64)
    s1 = _run(engine, gpu_device, *small)
    a = _run(engine, gpu_device, ff, fb, masks)
    b = _run(engine, gpu_device, ff, fb, masks)
    s2 = _run(engine, gpu_device, *small)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(s1[0], s2[0]) and np.array_equal(s1[1], s2[1])
    hole_f = np.broadcast_to(masks[:-1] > 0, a[0].shape)
    hole_b = np.broadcast_to(masks[1:] > 0, a[1].shape)
    assert np.array_equal(a[0][~hole_f], ff[~hole_f]) and np.array_equal(a[1][~hole_b], fb[~hole_b])
    assert np.isfinite(a[0]).all() and np.abs(a[0][hole_f]).max() > 0.1
