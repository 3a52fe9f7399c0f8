"""Recurrent flow completion without a GPU: the engine's plan replayed on the CPU
(tests/_replay_rfc.py) against oracle/rfc.py, which tests/test_oracle_golden.py pins to the reference module."""
import numpy as np
import pytest
import torch

import _replay_rfc as rr
from oracle.make_golden import rfc_inputs
from oracle.rfc import RfcOracle
from vsr_amd import _lib
from vsr_amd.engine import RfcEngine
from vsr_amd.synth import make_rfc_state_dict


@pytest.fixture(scope="module")
def rfc_sd():
    return make_rfc_state_dict(0)


@pytest.fixture(scope="module")
def host_engine(rfc_sd, built_lib):
    e = RfcEngine(rfc_sd, device=-1)
    yield e
    e.close()


@pytest.mark.parametrize("t,H,W", [(2, 64, 64), (6, 72, 104)])
def test_plan_replay_matches_oracle(host_engine, rfc_sd, t, H, W):
    view = rr.rfc_plan_view(_lib, host_engine, t, H, W)
    ff, fb, masks = rfc_inputs(5 + t, t, H, W)
    of, ob, _ = rr.replay_rfc(view, host_engine.packed_weights(), ff, fb, (masks[:, 0] > 0).astype(np.uint8))
    cf, cb, pf, pb = RfcOracle(rfc_sd).complete_bi(torch.from_numpy(ff), torch.from_numpy(fb), torch.from_numpy(masks))
    hole = masks[:-1] > 0
    assert np.abs(pf.numpy())[np.broadcast_to(hole, pf.shape)].max() > 0.1, "the network must predict something inside the hole"
    for name, got, ref in (("forward", of, cf.numpy()), ("backward", ob, cb.numpy())):
        err = np.abs(got - ref).max()
        assert err <= 2e-4, f"{name}: max abs err {err:.3e} (range {np.abs(ref).max():.1f})"
    # outside the hole the input flow passes through unchanged, bit for bit
    assert np.array_equal(of[np.broadcast_to(~hole, of.shape)], ff[np.broadcast_to(~hole, ff.shape)])
    assert view.flops == pytest.approx(host_engine.flops(t, H, W))
    view.close()


def test_schedule_shape(host_engine):
    """T steps per direction of the propagation; the deformable alignment is skipped on the first step of each direction."""
    t = 5
    view = rr.rfc_plan_view(_lib, host_engine, t, 64, 64)
    tags = [info.tag.decode() for info, _ in view.ops]
    T = t - 1
    assert tags.count("prop.bb0") == 2 * T and tags.count("prop.bb1") == 2 * T
    assert tags.count("prop.deform") == 2 * (T - 1) and tags.count("prop.deform.cols") == 2 * (T - 1)
    assert tags.count("prop.fusion") == 1 and tags.count("combine") == 1 and tags.count("enc.p3d.t") == 4
    view.close()


def test_strict_state_dict(rfc_sd, built_lib):
    bad = dict(rfc_sd)
    bad.pop("feat_prop_module.fusion.bias")
    with pytest.raises(_lib.VsrError, match="missing key"):
        RfcEngine(bad, device=-1)
    bad = dict(rfc_sd)
    bad["downsample.0.weight"] = np.zeros((32, 3, 1, 3, 3), np.float32)
    with pytest.raises(_lib.VsrError, match="shape mismatch"):
        RfcEngine(bad, device=-1)
