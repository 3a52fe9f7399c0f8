"""ProPainter generator without a GPU: plans replayed on the CPU (tests/_replay_pp.py) against
oracle/propainter.py, which tests/test_oracle_golden.py pins to the reference module."""
import numpy as np
import pytest
import torch

import _replay_pp as rp
from oracle.make_golden import propainter_inputs
from oracle.propainter import ProPainterOracle
from vsr_amd import _lib
from vsr_amd.synth import make_propainter_state_dict


@pytest.mark.parametrize("t,H,W", [(2, 32, 40), (6, 64, 96)])
def test_image_propagation_replay_matches_oracle(built_lib, t, H, W):
    frames, masks, ff, fb = propainter_inputs(50 + t, t, t, H, W)
    masked = frames * (1 - masks)
    view = rp.imgprop_plan_view(_lib, t, H, W)
    got, gm = rp.replay_imgprop(view, masked, ff, fb, masks[:, 0].astype(np.uint8))
    o = ProPainterOracle({})
    ref, rm = o.img_propagation(torch.from_numpy(masked), torch.from_numpy(ff), torch.from_numpy(fb), torch.from_numpy(masks).clone())
    assert np.array_equal(gm, rm[:, 0].numpy().astype(np.uint8))
    assert np.array_equal(got, ref.numpy()), "nearest-neighbour propagation copies pixels: bit-exact"
    if t > 1:
        assert gm.sum() < masks.sum(), "some hole pixels must get filled from neighbouring frames"
    view.close()


def _generator_case(seed, t, lt, H, W, sd):
    """inputs chained as PropainterInpaint.inpaint does (propainter_inpaint.py:283-341) + the oracle's output"""
    frames, masks, ff, fb = propainter_inputs(seed, t, lt, H, W)
    o = ProPainterOracle(sd)
    fr, mk = torch.from_numpy(frames), torch.from_numpy(masks)
    masked = fr * (1 - mk)
    prop, upd = o.img_propagation(masked[:lt], torch.from_numpy(ff), torch.from_numpy(fb), mk[:lt].clone())
    sel = torch.cat([fr[:lt] * (1 - mk[:lt]) + prop * mk[:lt], masked[lt:]])
    sel_upd = torch.cat([upd, mk[lt:]])
    ref = o.forward(sel, torch.from_numpy(ff), torch.from_numpy(fb), mk, sel_upd, lt)
    return sel.numpy(), ff, fb, masks[:, 0].astype(np.uint8), sel_upd[:, 0].numpy().astype(np.uint8), ref.numpy()


@pytest.fixture(scope="module")
def pp_sd():
    return make_propainter_state_dict(0)


@pytest.fixture(scope="module")
def pp_host_engine(pp_sd, built_lib):
    from vsr_amd.engine import PpEngine

    e = PpEngine(device=-1, state_dict=pp_sd)
    yield e
    e.close()


@pytest.mark.parametrize("t,lt,H,W,expect_unmasked", [(7, 5, 64, 96, False), (5, 3, 128, 192, True)])
def test_generator_replay_matches_oracle(pp_host_engine, pp_sd, t, lt, H, W, expect_unmasked):
    """Every packed weight (grouped / padded-channel convs, fused qkv), table, descriptor and elementwise op of
    InpaintGenerator.forward; the second case has attention windows outside the hole (per-frame attention) and a token
    grid that needs window padding."""
    sel, ff, fb, m_in, m_upd, ref = _generator_case(70 + t, t, lt, H, W, pp_sd)
    flags = pp_host_engine.window_flags(m_in[:lt])
    assert (0 in flags) == expect_unmasked and 1 in flags
    view = rp.gen_plan_view(_lib, pp_host_engine, t, lt, H, W, flags)
    out, _ = rp.replay_gen(view, pp_host_engine.packed_weights(), sel, ff, fb, m_in, m_upd, lt)
    err = np.abs(out - ref).max()
    assert err <= 5e-4, f"generator output: max abs err {err:.3e} (tanh range)"
    view.close()


@pytest.mark.parametrize("box", [(40, 60, 0, 0), (0, 9, 100, 192), (100, 128, 0, 30), (50, 70, 60, 130)])
def test_generator_replay_box(pp_host_engine, pp_sd, box):
    """vsr_pp_forward_box: with the promise that only a box of the output is read (the plugin blends a prediction in under the dilated
    mask), the soft composition's embedding and the decoder's convs run on the tokens / pixels that box depends on.  The replay starts
    from zeroed buffers and the elementwise ops keep whole images, so a range one row or column short anywhere in the chain (3x3
    convs, two align_corners upsamplings, the 7x7 / stride-3 fold) shows inside the box; inside it the output is the full plan's
    (up to torch-CPU's M-dependent matmul blocking), outside it differs, the FLOPs go down and the reference count stays."""
    import ctypes as C

    t, lt, H, W = 5, 3, 128, 192
    sel, ff, fb, m_in, m_upd, ref = _generator_case(75, t, lt, H, W, pp_sd)
    flags = pp_host_engine.window_flags(m_in[:lt])
    full = rp.gen_plan_view(_lib, pp_host_engine, t, lt, H, W, flags)
    want, _ = rp.replay_gen(full, pp_host_engine.packed_weights(), sel, ff, fb, m_in, m_upd, lt)
    full_flops = full.flops
    full.close()
    part = rp.gen_plan_view(_lib, pp_host_engine, t, lt, H, W, flags, box=box)
    try:
        got, _ = rp.replay_gen(part, pp_host_engine.packed_weights(), sel, ff, fb, m_in, m_upd, lt)
        y0, y1, x0, x1 = box[0], box[1], box[2], (box[3] if box[3] > box[2] else W)
        d = np.abs(got - want)
        assert d[:, :, y0:y1, x0:x1].max() <= 2e-5, d[:, :, y0:y1, x0:x1].max()
        assert np.abs(want[:, :, y0:y1, x0:x1] - ref[:, :, y0:y1, x0:x1]).max() <= 5e-4
        assert d.max() > 1e-2                               # ... and the ranges are ranges: far from the box nothing was computed
        assert part.flops < full_flops
        f = np.ascontiguousarray(flags, dtype=np.uint8)
        r = C.c_double()
        x = _lib.lib.vsr_pp_flops_box(pp_host_engine.handle, t, lt, H, W, f.ctypes.data_as(C.c_void_p), f.size, *box, C.byref(r))
        assert x == part.flops and abs(r.value - full_flops) <= 1e-6 * full_flops
    finally:
        part.close()


def test_generator_replay_encoder_cache(pp_host_engine, pp_sd):
    """vsr_pp_encode + vsr_pp_forward_cached: the encoder and the soft split are per-frame functions of a frame's inputs, and the
    plugin's windows overlap (a frame is a local frame of 2-3 windows and a reference frame of several more), so they run once per
    frame (PP_PLAN_ENCODE, in chunks of any size) and the generator (PP_PLAN_CACHED) starts from the cached features of its local
    frames and the cached tokens of its reference frames.  Same output as the full plan -- alone and together with a box promise --
    and the FLOPs of the two plans add up to the full plan's (nothing else changed)."""
    t, lt, H, W = 5, 3, 128, 192
    sel, ff, fb, m_in, m_upd, ref = _generator_case(75, t, lt, H, W, pp_sd)
    flags = pp_host_engine.window_flags(m_in[:lt])
    wts = pp_host_engine.packed_weights()
    full = rp.gen_plan_view(_lib, pp_host_engine, t, lt, H, W, flags)
    want, _ = rp.replay_gen(full, wts, sel, ff, fb, m_in, m_upd, lt)
    full_flops = full.flops
    full.close()
    # the frames are encoded in two calls, in an order that is not the window's: the reference frames with their tokens (only those
    # are read), the local ones without
    feats, toks = np.zeros((t, H // 4, W // 4, 128), np.float32), None
    enc_flops = 0.0
    for part, ntokf in (([4, 3], 2), ([1, 0, 2], 0)):
        ev = rp.gen_plan_view(_lib, pp_host_engine, len(part), ntokf, H, W, None, mode=1)
        f, k = rp.replay_encode(ev, wts, sel[part], m_in[part], m_upd[part])
        enc_flops += ev.flops
        ev.close()
        if toks is None:
            toks = np.zeros((t,) + k.shape[1:], np.float32)
        feats[part] = f
        toks[part[:ntokf]] = k[:ntokf]
    for box in (None, (50, 70, 60, 130)):
        cv = rp.gen_plan_view(_lib, pp_host_engine, t, lt, H, W, flags, box=box, mode=2)
        got, _ = rp.replay_gen(cv, wts, sel, ff, fb, m_in, m_upd, lt, cached=(feats, toks))
        y0, y1, x0, x1 = box if box else (0, H, 0, W)
        assert np.abs(got[:, :, y0:y1, x0:x1] - want[:, :, y0:y1, x0:x1]).max() <= 2e-5
        if box is None:
            assert abs(cv.flops + enc_flops - full_flops) <= 1e-9 * full_flops, (cv.flops, enc_flops, full_flops)
            assert not any(i.tag.decode().startswith("enc.") for i, _ in cv.ops)
        cv.close()


def test_generator_strict_state_dict(pp_sd, built_lib):
    from vsr_amd.engine import PpEngine

    bad = dict(pp_sd)
    bad.pop("sc.bias_conv.bias")
    with pytest.raises(_lib.VsrError, match="missing key"):
        PpEngine(device=-1, state_dict=bad)
    bad = dict(pp_sd)
    bad["encoder.layers.14.weight"] = np.zeros((256, 96, 3, 3), np.float32)
    with pytest.raises(_lib.VsrError, match="shape mismatch"):
        PpEngine(device=-1, state_dict=bad)


def test_generator_entry_points_have_no_cpu_path(pp_host_engine):
    """a handle that was packed without a device: every generator entry point -- the round-4 ones too -- fails with VSR_ERR_NOGPU"""
    import ctypes as C

    lib = _lib.lib
    f = np.zeros(64, np.float32)
    u = np.zeros(64, np.uint8)
    idx = np.zeros(2, np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    h = pp_host_engine.handle
    calls = [
        lambda: lib.vsr_pp_forward(h, P(f), P(f), P(f), P(u), P(u), 2, 2, 64, 96, P(u), 4, P(f), None),
        lambda: lib.vsr_pp_forward_box(h, P(f), P(f), P(f), P(u), P(u), 2, 2, 64, 96, P(u), 4, 8, 16, 8, 16, P(f), None),
        lambda: lib.vsr_pp_encode(h, P(f), P(u), P(u), 2, 1, 64, 96, P(f), P(f), None),
        lambda: lib.vsr_pp_forward_cached(h, P(f), P(f), P(idx), P(f), P(f), P(u), P(u), 2, 2, 64, 96, P(u), 4, 0, 0, 0, 0, P(f), None),
    ]
    for call in calls:
        assert call() == _lib.VSR_ERR_NOGPU and "no CPU fallback" in _lib.last_error()
    assert lib.vsr_pp_encode(h, P(f), P(u), P(u), 2, 3, 64, 96, P(f), P(f), None) == _lib.VSR_ERR_ARG          # more token frames than frames
    assert lib.vsr_pp_token_count(64, 96) == 6 * 8 and lib.vsr_pp_token_count(360, 1920) == 30 * 160
