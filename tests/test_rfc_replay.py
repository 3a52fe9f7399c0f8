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


def test_plan_replay_frame_groups(rfc_sd, built_lib):
    """A 68 / 70-frame batch of 1080p strips has operands beyond 2^31 elements (the stem's im2col, the full-resolution map in front of the
    last conv): RfcPlan::conv cuts such a conv into problems of consecutive frames, each with a 64-bit base and tables relative to its
    first frame.  VSR_RFC_SPAN_LIMIT (read once per process, hence a child) makes a small plan take that form: the replay still equals
    the oracle, the FLOPs are the same, and some ops carry several problems."""
    import json
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = f"""
import sys, json
sys.path.insert(0, {os.path.dirname(here)!r}); sys.path.insert(0, {here!r})
import numpy as np, torch
import _replay_rfc as rr
from oracle.make_golden import rfc_inputs
from oracle.rfc import RfcOracle
from vsr_amd import _lib
from vsr_amd.engine import RfcEngine
from vsr_amd.synth import make_rfc_state_dict
sd = make_rfc_state_dict(0)
e = RfcEngine(sd, device=-1)
t, H, W = 6, 72, 104
view = rr.rfc_plan_view(_lib, e, t, H, W)
ff, fb, masks = rfc_inputs(5 + t, t, H, W)
of, ob, _ = rr.replay_rfc(view, e.packed_weights(), ff, fb, (masks[:, 0] > 0).astype(np.uint8))
cf, cb, pf, pb = RfcOracle(sd).complete_bi(torch.from_numpy(ff), torch.from_numpy(fb), torch.from_numpy(masks))
print(json.dumps(dict(err=float(max(np.abs(of - cf.numpy()).max(), np.abs(ob - cb.numpy()).max())), flops=view.flops,
                      grouped=sorted(set(info.tag.decode() for info, _ in view.ops if info.kind == rr._replay.OP_GEMM and info.nitems > 1)),
                      nitems=max(info.nitems for info, _ in view.ops))))
"""
    res = {}
    for limit in ("0", "300000"):            # 0 = the default limit: one problem per conv at this size
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VSR_RFC_SPAN_LIMIT=limit), capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        res[limit] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["0"]["nitems"] == 1 and res["0"]["grouped"] == []
    assert res["300000"]["nitems"] > 1 and "enc.stem" in res["300000"]["grouped"], res["300000"]
    assert res["300000"]["err"] <= 2e-4 and res["0"]["err"] <= 2e-4
    assert res["300000"]["flops"] == pytest.approx(res["0"]["flops"])


def test_plan_builds_at_config4_batch_sizes(host_engine):
    """batch_generator(1200, 70) -> 68 / 44-frame batches of 1920 x 360 strips (and 70-frame ones for other clip lengths): the plans build
    (round 4's did not beyond 49 frames: "offset table entry exceeds int32") and the large convs are cut into frame groups"""
    import ctypes as C

    for t in (44, 68, 70):
        p = C.c_void_p()
        _lib.check(_lib.lib.vsr_rfc_plan_create(host_engine.handle, t, 360, 1920, C.byref(p)))
        assert _lib.lib.vsr_plan_flops(p) == pytest.approx(host_engine.flops(t, 360, 1920))
        _lib.lib.vsr_plan_destroy(p)
