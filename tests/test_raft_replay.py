"""RAFT without a GPU: the engine's plan -- packed / BatchNorm-folded weights, every offset table,
gather-GEMM descriptor and elementwise op -- replayed on the CPU (tests/_replay_raft.py) against oracle/raft.py, which
tests/test_oracle_golden.py pins to the reference module."""
import numpy as np
import pytest
import torch

import _replay_raft as rr
from oracle.raft import RaftOracle
from vsr_amd import _lib
from vsr_amd.engine import RaftEngine
from vsr_amd.synth import make_flow_frames, make_raft_state_dict


@pytest.fixture(scope="module")
def raft_sd():
    return make_raft_state_dict(0)


@pytest.fixture(scope="module")
def host_engine(raft_sd, built_lib):
    e = RaftEngine(raft_sd, device=-1)         # host only: packing + plan introspection
    yield e
    e.close()


@pytest.mark.parametrize("t,H,W,iters,tol", [(3, 128, 192, 3, 2e-4), (2, 136, 200, 20, 5e-3)])
def test_plan_replay_matches_oracle(host_engine, raft_sd, t, H, W, iters, tol):
    view = rr.raft_plan_view(_lib, host_engine, t, H, W, iters)
    frames = make_flow_frames(t, H, W, seed=3)
    fwd, bwd, _ = rr.replay_raft(view, host_engine.packed_weights(), frames)
    x = torch.from_numpy(frames).permute(0, 3, 1, 2).float().div(255) * 2 - 1
    of, ob = RaftOracle(raft_sd).flows_bi(x, iters)
    assert np.abs(of.numpy()).max() > 1.0, "the synthetic clip must produce a non-trivial flow"
    for name, got, ref in (("forward", fwd, of.numpy()), ("backward", bwd, ob.numpy())):
        err = np.abs(got - ref).max()
        assert err <= tol, f"{name} flow: max abs err {err:.3e} px (range {np.abs(ref).max():.1f} px)"
    assert view.flops == pytest.approx(host_engine.flops(t, H, W, iters))
    view.close()


def test_schedule_shape(host_engine):
    """One correlation problem per pair-direction, the upsampling mask only after the last iteration, fnet and cnet once per frame."""
    t, iters = 4, 5
    view = rr.raft_plan_view(_lib, host_engine, t, 128, 128, iters)
    tags = [info.tag.decode() for info, _ in view.ops]
    corr = [items for info, items in view.ops if info.tag == b"corr.volume"]
    assert len(corr) == 1 and len(corr[0]) == 2 * (t - 1)
    assert tags.count("upd.mask1") == 1 and tags.count("upd.mask2") == 1 and tags.count("flow.upsample") == 1
    assert tags.count("corr.lookup") == iters and tags.count("gru.zr") == 2 * iters and tags.count("flow.update") == iters
    assert tags.count("fnet.stem") == 1 and tags.count("cnet.stem") == 1
    stem = [items for info, items in view.ops if info.tag == b"fnet.stem"][0][0]
    assert stem.M == t * 64 * 64 and stem.N == 64 and stem.K == 160
    view.close()


def test_strict_state_dict(raft_sd, built_lib):
    bad = dict(raft_sd)
    bad.pop("update_block.gru.convq2.bias")
    with pytest.raises(_lib.VsrError, match="missing key"):
        RaftEngine(bad, device=-1)
    bad = dict(raft_sd)
    bad["fnet.norm1.weight"] = np.ones(64, np.float32)         # InstanceNorm2d has no parameters in this checkpoint
    with pytest.raises(_lib.VsrError, match="unexpected key"):
        RaftEngine(bad, device=-1)
    bad = dict(raft_sd)
    bad["update_block.encoder.convc1.weight"] = np.zeros((256, 320, 1, 1), np.float32)
    with pytest.raises(_lib.VsrError, match="shape mismatch"):
        RaftEngine(bad, device=-1)
    with pytest.raises(_lib.VsrError):                          # frame size must be a multiple of 8 and >= 128
        e = RaftEngine(raft_sd, device=-1)
        try:
            rr.raft_plan_view(_lib, e, 2, 100, 128, 2)
        finally:
            e.close()


def test_module_prefix_is_stripped(raft_sd, built_lib):
    e = RaftEngine({"module." + k: v for k, v in raft_sd.items()}, device=-1)      # DataParallel checkpoint layout
    assert e.packed_weights().size > 5_000_000
    e.close()


def test_no_cpu_fallback(raft_sd, built_lib):
    e = RaftEngine(raft_sd, device=-1)
    with pytest.raises(_lib.VsrError) as ei:
        import ctypes as C
        buf = np.zeros(16, np.uint8)
        out = np.zeros(16, np.float32)
        _lib.check(_lib.lib.vsr_raft_flows(e.handle, buf.ctypes.data_as(C.c_void_p), 2, 128, 128, 2, 0,
                                           out.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), None))
    assert ei.value.code == _lib.VSR_ERR_NOGPU
    e.close()
