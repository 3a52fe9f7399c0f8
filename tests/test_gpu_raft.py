"""RAFT on the MI355X: vsr_raft_flows through the C-ABI against oracle/raft.py (pinned to the
reference module by tests/test_oracle_golden.py).  fp32 end to end; the tolerance is absolute, in pixels of flow:
the 20-iteration recurrence amplifies summation-order differences (1e-6 on the input moves the flow by ~1e-3 px)."""
import numpy as np
import pytest
import torch

from oracle.raft import RaftOracle
from vsr_amd.engine import RaftEngine
from vsr_amd.synth import make_flow_frames, make_raft_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def raft_sd():
    return make_raft_state_dict(0)


@pytest.fixture(scope="module")
def engine(raft_sd, built_lib, gpu_device):
    e = RaftEngine(raft_sd, device=0)
    yield e
    e.close()


def _oracle_flows(sd, frames, iters):
    x = torch.from_numpy(frames).permute(0, 3, 1, 2).float().div(255) * 2 - 1
    return RaftOracle(sd).flows_bi(x, iters)


@pytest.mark.parametrize("t,H,W,iters,tol", [(3, 128, 192, 1, 2e-4), (3, 128, 192, 20, 5e-3), (2, 136, 200, 6, 1e-3),
                                              (3, 360, 640, 4, 1e-3)])
def test_flows_match_oracle(engine, raft_sd, gpu_device, t, H, W, iters, tol):
    frames = make_flow_frames(t, H, W, seed=5)
    fwd, bwd = engine.flows(torch.from_numpy(frames).to(gpu_device), iters=iters)
    torch.cuda.synchronize()
    of, ob = _oracle_flows(raft_sd, frames, iters)
    for name, got, ref in (("forward", fwd, of), ("backward", bwd, ob)):
        err = (got.cpu() - ref).abs().max().item()
        print(f"{name} {t}x{H}x{W} iters={iters}: max abs err {err:.3e} px, flow range {ref.abs().max().item():.1f} px")
        assert torch.isfinite(got).all()
        assert err <= tol, f"{name} flow: max abs err {err:.3e} px > {tol}"


def test_bgr_flag_and_determinism(engine, gpu_device):
    frames = make_flow_frames(3, 128, 160, seed=6)
    d = torch.from_numpy(frames).to(gpu_device)
    f1, b1 = engine.flows(d, iters=5)
    f2, b2 = engine.flows(d, iters=5)
    dbgr = torch.from_numpy(np.ascontiguousarray(frames[..., ::-1])).to(gpu_device)
    f3, b3 = engine.flows(dbgr, iters=5, bgr=True)
    torch.cuda.synchronize()
    assert torch.equal(f1, f2) and torch.equal(b1, b2)
    assert torch.equal(f1, f3) and torch.equal(b1, b3)


def test_time_reversal_property_at_strip_size(engine, gpu_device):
    """1080p strip (1920x360, the size ProPainter hands to RAFT): the backward flow of a clip is the forward flow of
    the reversed clip -- same frame pairs, other slot of the batch -- bit for bit; and the workspace survives the
    change of plan (small -> large -> small gives identical results)."""
    small = torch.from_numpy(make_flow_frames(2, 128, 128, seed=7)).to(gpu_device)
    s1 = engine.flows(small, iters=3)
    frames = make_flow_frames(3, 360, 1920, seed=8)
    d = torch.from_numpy(frames).to(gpu_device)
    fwd, bwd = engine.flows(d, iters=3)
    rf, rb = engine.flows(torch.flip(d, dims=[0]).contiguous(), iters=3)
    s2 = engine.flows(small, iters=3)
    torch.cuda.synchronize()
    assert torch.isfinite(fwd).all() and torch.isfinite(bwd).all()
    assert torch.equal(torch.flip(rb, dims=[0]), fwd) and torch.equal(torch.flip(rf, dims=[0]), bwd)
    assert torch.equal(s1[0], s2[0]) and torch.equal(s1[1], s2[1])


def test_frame_count_alternation_shares_one_clean_workspace(engine, raft_sd, built_lib, gpu_device):
    """The runs of a ProPainter batch alternate between frame counts at one frame size (raft_runs: 18, 18, 18, 17 frames of 68): the
    workspace is cleared by extent, not on every change of t (flow_engine.hip ensure_clean).  4 -> 3 -> 4 -> 2 frames on one engine give
    what a fresh engine gives for each clip, bit for bit -- a shorter run leaves the longer run's frames behind it and must not see them."""
    clips = [make_flow_frames(t, 136, 200, seed=20 + i) for i, t in enumerate((4, 3, 4, 2))]
    got = [engine.flows(torch.from_numpy(c).to(gpu_device), iters=4) for c in clips]
    torch.cuda.synchronize()
    for c, (f, b) in zip(clips, got):
        fresh = RaftEngine(raft_sd, device=0)
        ff, fb = fresh.flows(torch.from_numpy(c).to(gpu_device), iters=4)
        torch.cuda.synchronize()
        assert torch.equal(f, ff) and torch.equal(b, fb)
        fresh.close()
