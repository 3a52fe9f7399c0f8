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
    # forward pair-directions by GEMM; the backward ones are their transposes (one pass right behind the GEMMs)
    assert len(corr) == 1 and len(corr[0]) == t - 1 and tags[tags.index("corr.volume") + 1] == "corr.transpose"
    assert tags.count("upd.mask1") == 1 and tags.count("upd.mask2") == 1 and tags.count("flow.upsample") == 1
    assert tags.count("corr.lookup") == iters and tags.count("gru.zr") == 2 * iters and tags.count("flow.update") == iters
    assert tags.count("fnet.stem") == 1 and tags.count("cnet.stem") == 1
    stem = [items for info, items in view.ops if info.tag == b"fnet.stem"][0][0]
    assert stem.M == t * 64 * 64 and stem.N == 64 and stem.K == 160
    view.close()


def test_context_share_of_the_gru_is_computed_once(host_engine, raft_sd):
    """conv(cat[h, inp, motion]) = conv_inp(inp) + bias + conv_rest(cat[h, motion]) and inp never changes over the iterations
    (raft.py:114-116,127-133): the plan holds four context products (z|r and q of both GRU passes, K = 5 x 128) in front of the loop,
    the per-iteration GRU convs contract K = 5 x 256 and take the context product as their residual -- and the plan without the hoist
    (VSR_RAFT_CTX_HOIST=0, a fresh process: the library reads the switch once) replays to the same flows within fp32 rounding."""
    import json
    import os
    import subprocess
    import sys

    t, H, W, iters = 2, 128, 160, 4
    view = rr.raft_plan_view(_lib, host_engine, t, H, W, iters)
    ops = [(info.tag.decode(), items) for info, items in view.ops]
    tags = [tg for tg, _ in ops]
    assert tags.count("gru.zr.ctx") == 2 and tags.count("gru.q.ctx") == 2
    assert max(tags.index("gru.zr.ctx"), tags.index("gru.q.ctx")) < tags.index("corr.lookup")          # in front of the first iteration
    for tg, items in ops:
        if tg in ("gru.zr", "gru.q"):
            assert items[0].K == 5 * 256 and items[0].bufR >= 0 and items[0].offBias < 0
        if tg in ("gru.zr.ctx", "gru.q.ctx"):
            assert items[0].K == 5 * 128 and items[0].bufR < 0 and items[0].offBias >= 0
    frames = make_flow_frames(t, H, W, seed=9)
    fwd, bwd, _ = rr.replay_raft(view, host_engine.packed_weights(), frames)
    flops = view.flops
    view.close()
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\nimport _replay_raft as rr\nfrom vsr_amd import _lib\nfrom vsr_amd.engine import RaftEngine\n"
            "from vsr_amd.synth import make_flow_frames, make_raft_state_dict\n"
            "e = RaftEngine(make_raft_state_dict(0), device=-1)\nview = rr.raft_plan_view(_lib, e, %d, %d, %d, %d)\n"
            "tags = [info.tag.decode() for info, _ in view.ops]\n"
            "f, b, _ = rr.replay_raft(view, e.packed_weights(), make_flow_frames(%d, %d, %d, seed=9))\n"
            "np.save(sys.argv[1], np.stack([f, b]))\nprint(json.dumps({'ctx': tags.count('gru.zr.ctx'), 'flops': view.flops}))\n"
            ) % (os.path.dirname(here), here, t, H, W, iters, t, H, W)
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"raft_nohoist_{os.getpid()}.npy")
    r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, VSR_RAFT_CTX_HOIST="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    plain = json.loads(r.stdout.strip().splitlines()[-1])
    ref = np.load(out)
    os.remove(out)
    assert plain["ctx"] == 0 and flops < plain["flops"]
    err = max(np.abs(fwd - ref[0]).max(), np.abs(bwd - ref[1]).max())
    assert err <= 2e-4, f"hoisted vs plain plan: {err:.3e} px"


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
