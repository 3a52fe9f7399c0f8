"""ProPainter generator stages on the MI355X through the C-ABI against oracle/propainter.py."""
import os

import numpy as np
import pytest
import torch

from vsr_amd import switches
from oracle.make_golden import propainter_inputs
from oracle.propainter import ProPainterOracle
from vsr_amd.engine import PpEngine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(built_lib, gpu_device):
    e = PpEngine(device=0)
    yield e
    e.close()


@pytest.mark.parametrize("t,H,W", [(2, 32, 40), (6, 64, 96), (5, 360, 640)])
def test_image_propagation_bit_exact(engine, gpu_device, t, H, W):
    frames, masks, ff, fb = propainter_inputs(60 + t, t, t, H, W)
    masked = frames * (1 - masks)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
    got, gm = engine.img_propagation(d(masked), d(ff), d(fb), d(masks[:, 0].astype(np.uint8)))
    torch.cuda.synchronize()
    ref, rm = ProPainterOracle({}).img_propagation(torch.from_numpy(masked), torch.from_numpy(ff), torch.from_numpy(fb),
                                                   torch.from_numpy(masks).clone())
    mism = (gm.cpu().numpy() != rm[:, 0].numpy().astype(np.uint8)).mean()
    diff = (got.cpu() != ref).float().mean().item()
    print(f"imgprop {t}x{H}x{W}: mask mismatches {mism:.2e}, pixel mismatches {diff:.2e}, filled {1 - rm.sum().item() / masks.sum():.3f}")
    # thresholded decisions (flow consistency, nearest rounding) can flip on last-bit differences of the sampling coordinates
    assert mism <= 1e-4 and diff <= 1e-4


@pytest.fixture(scope="module")
def pp_sd():
    from vsr_amd.synth import make_propainter_state_dict

    return make_propainter_state_dict(0)


@pytest.fixture(scope="module")
def gen_engine(pp_sd, built_lib, gpu_device):
    e = PpEngine(device=0, state_dict=pp_sd)
    yield e
    e.close()


def _generator_case(seed, t, lt, H, W, sd):
    frames, masks, ff, fb = propainter_inputs(seed, t, lt, H, W)
    o = ProPainterOracle(sd)
    fr, mk = torch.from_numpy(frames), torch.from_numpy(masks)
    masked = fr * (1 - mk)
    prop, upd = o.img_propagation(masked[:lt], torch.from_numpy(ff), torch.from_numpy(fb), mk[:lt].clone())
    sel = torch.cat([fr[:lt] * (1 - mk[:lt]) + prop * mk[:lt], masked[lt:]])
    sel_upd = torch.cat([upd, mk[lt:]])
    ref = o.forward(sel, torch.from_numpy(ff), torch.from_numpy(fb), mk, sel_upd, lt)
    return sel.numpy(), ff, fb, masks[:, 0].astype(np.uint8), sel_upd[:, 0].numpy().astype(np.uint8), ref.numpy()


@pytest.mark.parametrize("t,lt,H,W", [(7, 5, 64, 96), (5, 3, 128, 192), (6, 4, 240, 432)])
def test_generator_matches_oracle(gen_engine, pp_sd, gpu_device, t, lt, H, W):
    sel, ff, fb, m_in, m_upd, ref = _generator_case(80 + t, t, lt, H, W, pp_sd)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
    out = gen_engine.forward(d(sel), d(ff), d(fb), d(m_in), d(m_upd), lt)
    torch.cuda.synchronize()
    err = np.abs(out.cpu().numpy() - ref).max()
    print(f"generator {t}/{lt} x {H}x{W}: max abs err {err:.3e} (tanh output)")
    assert torch.isfinite(out).all()
    assert err <= 2e-3


@pytest.mark.parametrize("t,lt,H,W,precision", [(7, 5, 64, 96, "f32"), (6, 4, 240, 432, "f32"), (6, 4, 240, 432, "f16")])
def test_fused_window_attention_equals_three_ops(built_lib, gpu_device, tmp_path, t, lt, H, W, precision):
    """VSR_PP_FLASH=1 (default: ONE fused launch per block, online softmax, no score matrix -- pp_attn_kernels.hip) against
    VSR_PP_FLASH=0 (the plan's three ops QK^T / k_softmax_rows / P.V) on the same inputs: the same tanh output up to fp32 summation
    order in the exact mode; in the f16 mode both run on fp16 operands (k_pp_flash_attn_f16 keeps the softmax and the running output
    in fp32) and differ by that mode's own rounding."""
    import subprocess
    import sys

    outs = {}
    for v in ("0", "1"):
        path = str(tmp_path / f"flash{v}.npy")
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_pp_flash_child.py"), str(t), str(lt), str(H), str(W), precision, path],
                           env=dict(os.environ, VSR_PP_FLASH=v), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
        outs[v] = np.load(path)
    err = float(np.abs(outs["0"] - outs["1"]).max())
    print(f"fused vs three-op window attention {t}/{lt} x {H}x{W} [{precision}]: max |d| {err:.3e} (tanh output)")
    assert np.isfinite(outs["1"]).all()
    assert err <= (2e-5 if precision == "f32" else 2e-2)


def test_generator_strip_size_smoke(gen_engine, gpu_device):
    """1080p strip (1920x360), 11 local + 4 reference frames: finite, deterministic, survives a plan change."""
    t, lt, H, W = 15, 11, 360, 1920
    frames, masks, ff, fb = propainter_inputs(91, t, lt, H, W)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
    m8 = masks[:, 0].astype(np.uint8)
    args = (d(frames * (1 - masks)), d(ff), d(fb), d(m8), d(m8), lt)
    a = gen_engine.forward(*args)
    small = propainter_inputs(92, 3, 2, 64, 96)
    gen_engine.forward(d(small[0]), d(small[2]), d(small[3]), d(small[1][:, 0].astype(np.uint8)), d(small[1][:, 0].astype(np.uint8)), 2)
    b = gen_engine.forward(*args)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and a.abs().max() <= 1.0
    assert torch.equal(a, b)


@pytest.mark.skipif(not switches.on("VSR_PP_DECODE_BOX"),
                    reason="the generator's decoder box is switched off (VSR_PP_DECODE_BOX=0; default on since round 5)")
@pytest.mark.parametrize("box", [(224, 360, 0, 0), (224, 360, 280, 1640), (0, 64, 0, 512), (120, 200, 1400, 1920)])
def test_generator_decoder_box(gen_engine, gpu_device, box):
    """vsr_pp_forward_box at the 1080p strip size: inside the promised box the output is the one of the call without a promise bit
    for bit (the same products in the same order per output element), also after the stale rows of an earlier, different box."""
    t, lt, H, W = 15, 11, 360, 1920
    frames, masks, ff, fb = propainter_inputs(93, t, lt, H, W)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
    m8 = masks[:, 0].astype(np.uint8)
    args = (d(frames * (1 - masks)), d(ff), d(fb), d(m8), d(m8), lt)
    want = gen_engine.forward(*args).clone()
    gen_engine.forward(*args, box=(8, 40, 8, 200))                       # leaves other rows stale
    got = gen_engine.forward(*args, box=box)
    torch.cuda.synchronize()
    y0, y1, x0, x1 = box[0], box[1], box[2], (box[3] if box[3] > box[2] else W)
    assert torch.equal(got[:, :, y0:y1, x0:x1], want[:, :, y0:y1, x0:x1])
    assert torch.isfinite(got).all()


@pytest.mark.skipif(not switches.on("VSR_PP_ENC_CACHE"),
                    reason="the per-frame encoder cache is switched off (VSR_PP_ENC_CACHE=0; default on since round 5)")
@pytest.mark.parametrize("t,lt,H,W", [(5, 3, 128, 192), (15, 11, 360, 1920)])
def test_generator_encoder_cache(gen_engine, gpu_device, t, lt, H, W):
    """vsr_pp_encode + vsr_pp_forward_cached: the frames encoded once, in two calls and in another order than the window's, give the
    output of vsr_pp_forward bit for bit (same GEMM rows, same K order) -- alone and with a box promise."""
    frames, masks, ff, fb = propainter_inputs(95, t, lt, H, W)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
    m8 = masks[:, 0].astype(np.uint8)
    sel = d(frames * (1 - masks))
    dm = d(m8)
    want = gen_engine.forward(sel, d(ff), d(fb), dm, dm, lt).clone()
    refs, local = list(range(t - 1, lt - 1, -1)), list(range(lt))
    f1, k1 = gen_engine.encode(sel[refs].contiguous(), dm[refs].contiguous(), dm[refs].contiguous(), len(refs))
    f2, _ = gen_engine.encode(sel[local].contiguous(), dm[local].contiguous(), dm[local].contiguous(), 0)
    feat = torch.cat([f1, f2])
    idx = [len(refs) + k for k in range(lt)] + [refs.index(k) for k in range(lt, t)]
    flags = gen_engine.window_flags(m8[:lt])
    got = gen_engine.forward_cached(feat, k1, idx, d(ff), d(fb), dm, dm, lt, H, W, flags)
    box = (H // 2 // 8 * 8, H, W // 4 // 8 * 8, W // 4 * 3 // 8 * 8)
    got_box = gen_engine.forward_cached(feat, k1, idx, d(ff), d(fb), dm, dm, lt, H, W, flags, box=box)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(got_box[:, :, box[0]:box[1], box[2]:box[3]], want[:, :, box[0]:box[1], box[2]:box[3]])


def test_propainter_plugin_matches_oracle(built_lib, gpu_device, pp_sd):
    """PropainterInpaint.__call__ (row a13) end to end -- crop, mask dilation, RAFT, flow completion, image propagation,
    sliding neighbour / reference windows of the generator, u8 overlap blending -- against the restated reference wrapper
    built from the three CPU oracles.  RAFT runs 3 iterations on both sides to bound the CPU time."""
    from oracle.propainter_wrapper import PropainterOracle
    from oracle.raft import RaftOracle
    from oracle.rfc import RfcOracle
    from vsr_amd.backend.inpaint.propainter_inpaint import PropainterInpaint
    from vsr_amd.backend.tools.inpaint_tools import create_mask
    from vsr_amd.synth import make_clip, make_raft_state_dict, make_rfc_state_dict

    H, W, n = 288, 704, 7
    box = (236, 268, 120, 600)                                      # ymin, ymax, xmin, xmax
    frames = list(make_clip(n, H, W, box, seed=5))
    mask = create_mask((H, W), [(box[2], box[3], box[0], box[1])])
    sds = {"raft": make_raft_state_dict(0), "rfc": make_rfc_state_dict(0), "propainter": pp_sd}
    plug = PropainterInpaint("cuda:0", sds)
    plug.raft_iter = 3
    got = plug(frames, mask)
    plug.close()
    ora = PropainterOracle(RaftOracle(sds["raft"]), RfcOracle(sds["rfc"]), ProPainterOracle(pp_sd), raft_iter=3)
    ref = ora(frames, mask)
    assert len(got) == n
    g, r = np.stack(got).astype(np.float64), np.stack(ref).astype(np.float64)
    changed = (np.stack(ref) != np.stack(frames)).any(axis=(0, 3))
    assert changed.any(), "the plugin must repaint the masked strip"
    assert np.array_equal(np.stack(got)[:, ~changed], np.stack(frames)[:, ~changed]), "pixels the reference leaves alone stay bit-identical"
    mse = ((g - r)[:, changed] ** 2).mean()
    psnr = 99.0 if mse == 0 else 20 * np.log10(255.0 / np.sqrt(mse))
    dmax = np.abs(g - r).max()
    print(f"propainter plugin: PSNR vs oracle on repainted pixels {psnr:.1f} dB, max |d| {dmax:.0f}, repainted {changed.mean():.3f} of the frame")
    assert psnr >= 50.0


@pytest.mark.parametrize("precision,lanes", [("f32", 2), ("f32", 3), ("f16", 2)])
def test_generator_window_lanes_give_the_same_frames(built_lib, gpu_device, pp_sd, precision, lanes):
    """VSR_PP_LANES / PropainterInpaint.gen_lanes: the sliding windows of a call alternate over generator instances on their own
    streams, the blends chained in window order -- the frames are those of the single-stream loop bit for bit, repeatedly (a race
    would show as run-to-run differences); 23 frames = five windows, reference frames included.  In the guarded arithmetics ("f16")
    every lane has a host thread of its own (each generator call ends with a read of the range flag)."""
    from vsr_amd.backend.inpaint.propainter_inpaint import PropainterInpaint
    from vsr_amd.backend.tools.inpaint_tools import create_mask
    from vsr_amd.synth import make_clip, make_raft_state_dict, make_rfc_state_dict

    H, W, n = 288, 704, 23
    box = (236, 268, 120, 600)
    frames = list(make_clip(n, H, W, box, seed=8))
    mask = create_mask((H, W), [(box[2], box[3], box[0], box[1])])
    sds = {"raft": make_raft_state_dict(0), "rfc": make_rfc_state_dict(0), "propainter": pp_sd}
    plug = PropainterInpaint("cuda:0", sds, precision=precision)
    plug.raft_iter = 3
    plug.gen_lanes = 1
    want = np.stack(plug(frames, mask))
    plug.gen_lanes = lanes
    for _ in range(3):
        got = np.stack(plug(frames, mask))
        assert np.array_equal(got, want)
    assert len(plug._lane_models) == lanes - 1
    plug.close()
    assert (want != np.stack(frames)).any()


def test_raft_run_lanes_give_the_same_frames(built_lib, gpu_device, pp_sd):
    """VSR_RAFT_LANES / PropainterInpaint.raft_lanes: the runs of consecutive pairs a call's RAFT pass is cut into alternate over RAFT
    instances on their own streams; pairs are independent, so the frames are those of one run on one stream, bit for bit"""
    from vsr_amd.backend.inpaint.propainter_inpaint import PropainterInpaint
    from vsr_amd.backend.tools.inpaint_tools import create_mask
    from vsr_amd.synth import make_clip, make_raft_state_dict, make_rfc_state_dict

    H, W, n = 288, 704, 23
    box = (236, 268, 120, 600)
    frames = list(make_clip(n, H, W, box, seed=9))
    mask = create_mask((H, W), [(box[2], box[3], box[0], box[1])])
    sds = {"raft": make_raft_state_dict(0), "rfc": make_rfc_state_dict(0), "propainter": pp_sd}
    plug = PropainterInpaint("cuda:0", sds)
    plug.raft_iter = 3
    plug.raft_lanes, plug.raft_max_pairs = 1, 64                        # one run of 22 pairs
    want = np.stack(plug(frames, mask))
    plug.raft_lanes, plug.raft_max_pairs = 2, 12                        # four runs of <= 6 pairs on two lanes
    for _ in range(3):
        assert np.array_equal(np.stack(plug(frames, mask)), want)
    assert len(plug._lane_rafts) == 1
    plug.raft_lanes = 1                                                 # ... and the same runs on one lane
    assert np.array_equal(np.stack(plug(frames, mask)), want)
    plug.close()
