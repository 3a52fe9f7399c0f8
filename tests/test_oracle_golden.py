"""Pin the oracle against fixtures generated from the reference's own modules (oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sttn_auto
from oracle.sttn_net import SttnNet
from vsr_amd.synth import make_state_dict, state_dict_spec

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 2e-4   # the fixtures were made on another CPU: oneDNN/MKL kernels differ in summation order


def _close(a, b, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max() / max(1e-6, np.abs(b).max())
    assert err < RTOL, f"{what}: rel err {err:.3e}"


def test_state_dict_spec_matches_reference_counts():
    spec = state_dict_spec("auto")
    assert len(spec) == 112
    assert sum(int(np.prod(s)) for _, s in spec) == 16556163


def test_auto_net_matches_reference():
    g = np.load(os.path.join(GOLD, "sttn_auto_net.npz"))
    net = SttnNet(make_state_dict(0, "auto"), "auto")
    frames = np.random.default_rng(int(g["frames_seed"])).integers(0, 256, size=(3, 120, 640, 3), dtype=np.uint8)
    x = sttn_auto.STTNInpaintOracle.to_tensors(list(frames)) * 2 - 1
    with torch.no_grad():
        feat = net.encoder(x)
        pred = net.infer(feat)
        out = torch.tanh(net.decoder(pred[:2]))
    _close(feat.reshape(-1).numpy()[g["feat_idx"]], g["feat_val"], "encoder samples")
    _close(pred.reshape(-1).numpy()[g["pred_idx"]], g["pred_val"], "infer samples")
    _close(out[:, :, ::2, ::4].numpy(), g["out_sub"], "decoder output")
    for name, t in (("feat", feat), ("pred", pred), ("out", out)):
        _close(float((t.double() ** 2).sum()), float(g[name + "_sq"]), name + " energy")


def test_det_net_matches_reference_and_ignores_mask():
    g = np.load(os.path.join(GOLD, "sttn_det_net.npz"))
    net = SttnNet(make_state_dict(1, "det"), "det")
    rng = np.random.default_rng(int(g["x_seed"]))
    x = torch.from_numpy(rng.random((2, 3, 240, 432), dtype=np.float32) * 2 - 1)
    with torch.no_grad():
        pred = net.infer(net.encoder(x))      # the reference was given a mask; it must not matter
        out = torch.tanh(net.decoder(pred[:1]))
    _close(pred.reshape(-1).numpy()[g["pred_idx"]], g["pred_val"], "det infer samples")
    _close(out[:, :, ::4, ::4].numpy(), g["out_sub"], "det decoder output")


def test_batch_generator_matches_reference():
    gold = json.load(open(os.path.join(GOLD, "batch_generator.json")))
    for key, sizes in gold.items():
        n, m = (int(v) for v in key.split(","))
        assert [len(b) for b in sttn_auto.batch_generator(list(range(n)), m)] == sizes, key
    # SURVEY.md 8(a) a19 spot values
    assert gold["1200,50"] == [47] * 25 + [25]
    assert gold["300,50"] == [46] * 6 + [24]


def test_window_schedule_L50():
    """Appendix A of SURVEY.md: (6,4),(11,3|4)...(10,4); 104 decodes; last 4 frames decoded once."""
    o = sttn_auto.STTNInpaintOracle.__new__(sttn_auto.STTNInpaintOracle)
    o.neighbor_stride, o.ref_length = 5, 10
    sched = o.window_schedule(50)
    assert len(sched) == 10
    assert (len(sched[0][0]), len(sched[0][1])) == (6, 4)
    assert (len(sched[-1][0]), len(sched[-1][1])) == (10, 4)
    assert sum(len(n) for n, _ in sched) == 104 and sum(len(n) + len(r) for n, r in sched) == 140
    visits = np.zeros(50, int)
    for n, _ in sched:
        visits[n] += 1
    assert list(np.nonzero(visits == 1)[0]) == [46, 47, 48, 49]
    assert list(np.nonzero(visits == 3)[0]) == list(range(5, 45, 5))


def test_raft_matches_reference():
    """oracle/raft.py against the reference RAFT module (fixture from oracle/make_golden.py): encoders, 1 and 20
    update iterations, both directions, convex upsampling.  20 recurrent iterations amplify last-bit differences
    between CPUs (measured: a 1e-6 input perturbation moves the flow by 8e-4 px), hence the absolute tolerance."""
    from oracle.raft import RaftOracle
    from vsr_amd.synth import make_flow_frames, make_raft_state_dict, raft_state_dict_spec

    spec = raft_state_dict_spec()
    assert sum(int(np.prod(s)) for k, s in spec if k.endswith(("weight", "bias")) and "norm3" not in k) == 5257536
    g = np.load(os.path.join(GOLD, "raft.npz"))
    o = RaftOracle(make_raft_state_dict(0))
    frames = make_flow_frames(3, 128, 192, seed=int(g["frames_seed"]))
    x = torch.from_numpy(frames).permute(0, 3, 1, 2).float().div(255) * 2 - 1
    with torch.no_grad():
        _close(o.encoder(x[:1], "fnet.", "instance")[:, ::8].numpy(), g["fmap_sub"], "fnet")
        _close(o.encoder(x[:1], "cnet.", "batch")[:, ::8].numpy(), g["cmap_sub"], "cnet")
    for iters, tol in ((1, 1e-4), (20, 2e-2)):
        lo_f, up_f = o.forward(x[:-1], x[1:], iters)
        lo_b, up_b = o.forward(x[1:], x[:-1], iters)
        for name, t in (("low_f", lo_f), ("up_f", up_f), ("low_b", lo_b), ("up_b", up_b)):
            ref = g[f"{name}_{iters}"]
            t = t[..., ::2, ::3] if name.startswith("up") else t          # the fixture keeps every 2nd row / 3rd column
            err = np.abs(t.numpy() - ref).max()
            assert err <= tol, f"{name} after {iters} iterations: max abs err {err:.3e} px (flow range {np.abs(ref).max():.1f})"


def test_rfc_matches_reference():
    """oracle/rfc.py against the reference RecurrentFlowCompleteNet (fixture from oracle/make_golden.py, the reference
    run with oracle/deform_conv.py standing in for the absent torchvision.ops.deform_conv2d)."""
    from oracle.make_golden import rfc_inputs
    from oracle.rfc import RfcOracle
    from vsr_amd.synth import make_rfc_state_dict, rfc_state_dict_spec

    assert sum(int(np.prod(s)) for _, s in rfc_state_dict_spec()) == 5079555
    g = np.load(os.path.join(GOLD, "rfc.npz"))
    ff, fb, masks = rfc_inputs(int(g["seed"]), 5, 64, 96)
    cf, cb, pf, pb = RfcOracle(make_rfc_state_dict(0)).complete_bi(torch.from_numpy(ff), torch.from_numpy(fb), torch.from_numpy(masks))
    _close(pf.numpy(), g["pred_f"], "predicted forward flows")
    _close(pb.numpy(), g["pred_b"], "predicted backward flows")
    _close(cf[:, :, ::2, ::2].numpy(), g["comb_f"], "combined forward flows")
    _close(cb[:, :, ::2, ::2].numpy(), g["comb_b"], "combined backward flows")


def test_deform_conv_restatement_reduces_to_conv2d():
    """zero offsets + unit mask: deform_conv2d is a plain conv; integer offsets shift the sampling grid (zero outside)."""
    from oracle.deform_conv import deform_conv2d

    rng = np.random.default_rng(4)
    x = torch.from_numpy(rng.standard_normal((2, 32, 9, 11)).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal((8, 32, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(8).astype(np.float32))
    off = torch.zeros(2, 2 * 9 * 16, 9, 11)
    ref = torch.nn.functional.conv2d(x, w, b, padding=1)
    assert torch.allclose(deform_conv2d(x, off, w, b, 1, 1, 1, torch.ones(2, 9 * 16, 9, 11)), ref, atol=2e-4)
    off2 = off.clone()
    off2[:, 1::2] = 1.0                                    # every tap samples one pixel to the right
    xs = torch.nn.functional.pad(x, (0, 1))[..., 1:]       # shift left, zero fill
    # (the first output column differs by construction: the shifted conv pads where the deformable one still sees x = 0)
    assert torch.allclose(deform_conv2d(x, off2, w, b, 1, 1, 1, torch.ones(2, 9 * 16, 9, 11))[..., 1:],
                          torch.nn.functional.conv2d(xs, w, b, padding=1)[..., 1:], atol=2e-4)


def test_propainter_generator_matches_reference():
    """oracle/propainter.py (image propagation + generator forward) against the reference InpaintGenerator (fixture from
    oracle/make_golden.py; deform_conv2d restated)."""
    from oracle.make_golden import propainter_inputs
    from oracle.propainter import ProPainterOracle
    from vsr_amd.synth import make_propainter_state_dict, propainter_state_dict_spec

    assert sum(int(np.prod(s)) for k, s in propainter_state_dict_spec() if not k.endswith("valid_ind_rolled")) == 39429667
    g = np.load(os.path.join(GOLD, "propainter.npz"))
    t, lt, h, w = 7, 5, 64, 96
    frames, masks, ff, fb = (torch.from_numpy(a) for a in propainter_inputs(int(g["seed"]), t, lt, h, w))
    o = ProPainterOracle(make_propainter_state_dict(0))
    masked = frames * (1 - masks)
    prop, upd = o.img_propagation(masked[:lt], ff, fb, masks[:lt].clone())
    _close(prop[:, :, ::2, ::2].numpy(), g["prop"], "propagated frames")
    assert np.array_equal(upd[:, 0].numpy().astype(np.uint8), g["upd_mask"])
    assert upd.sum() < masks[:lt].sum(), "image propagation must fill part of the hole"
    upd_frames = frames[:lt] * (1 - masks[:lt]) + prop * masks[:lt]
    out = o.forward(torch.cat([upd_frames, masked[lt:]]), ff, fb, masks, torch.cat([upd, masks[lt:]]), lt)
    err = np.abs(out.numpy() - g["out"]).max()
    assert err <= 2e-3, f"generator output: max abs err {err:.3e} (tanh range)"


@pytest.mark.parametrize("sh,sw,dh,dw", [(360, 1920, 120, 640), (120, 640, 360, 1920), (240, 1280, 120, 640), (120, 640, 240, 1280),
                                         (159, 853, 120, 640), (120, 640, 159, 853), (720, 3840, 120, 640), (7, 5, 3, 11)])
def test_cv2_resize_restatement_agrees_with_an_independent_bilinear(sh, sw, dh, dw):
    """opencv is absent, so cv2.resize stays "parity unpinned" -- but its sampling rule (half-pixel centres, source index clamped at
    the borders, no antialiasing) is also torch.nn.functional.interpolate(bilinear, align_corners=False), written by other people:
    the float path of the restatement must agree with it to rounding, the uint8 fixed-point path (11-bit coefficients, 22-bit
    product shift) within one level of its rounded values.  Catches any error in the coordinate mapping or the border rule."""
    import numpy as np
    import torch

    from oracle import cv2_restate as cv2r

    rng = np.random.default_rng(sh * 7 + dw)
    img = rng.integers(0, 256, size=(sh, sw, 3), dtype=np.uint8)
    ref = torch.nn.functional.interpolate(torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None].double(), size=(dh, dw),
                                          mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    got_f = cv2r.resize_linear(img.astype(np.float32), (dw, dh))
    assert got_f.shape == (dh, dw, 3) and got_f.dtype == np.float32
    # OpenCV takes the source coordinate in fp32: at x ~ 1900 its fraction is good to ~6e-5, times a pixel step of up to 255
    assert np.abs(got_f - ref).max() <= 0.04
    got_u = cv2r.resize_linear(img, (dw, dh))
    assert got_u.dtype == np.uint8
    d = np.abs(got_u.astype(np.float64) - ref)
    assert d.max() <= 1.0 + 0.04 and (d > 0.75).mean() < 0.02


def test_deform_conv_restatement_against_grid_sample():
    """torchvision is absent, so deform_conv2d stays unpinned against its own binary -- but its definition (Dai et al.; the layout
    torchvision documents: offsets [B, 2*G*K, H, W] ordered (group, tap, (dy, dx)), modulation mask [B, G*K, H, W]) can be composed
    from torch's own bilinear sampler, written by other people: per offset group and tap, the group's channels sampled at
    (y + ky - 1 + dy, x + kx - 1 + dx) with zeros outside (grid_sample, align_corners=True on pixel coordinates), times the mask,
    contracted with that tap's weights.  Fractional offsets, different per group and tap, spatially varying."""
    from oracle.deform_conv import deform_conv2d

    rng = np.random.default_rng(11)
    B, C, H, W, G, Co = 2, 32, 10, 13, 4, 6
    K = 9
    x = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal((Co, C, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(Co).astype(np.float32))
    off = torch.from_numpy((rng.standard_normal((B, 2 * G * K, H, W)) * 1.7).astype(np.float32))
    mask = torch.from_numpy(rng.random((B, G * K, H, W)).astype(np.float32))
    got = deform_conv2d(x, off, w, b, 1, 1, 1, mask)

    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    want = b.double()[None, :, None, None].expand(B, Co, H, W).clone()
    cg = C // G
    for g in range(G):
        xg = x[:, g * cg:(g + 1) * cg].double()
        for t in range(K):
            ky, kx = t // 3, t % 3
            dy = off[:, g * 2 * K + 2 * t].double()
            dx = off[:, g * 2 * K + 2 * t + 1].double()
            py, px = ys[None] + ky - 1 + dy, xs[None] + kx - 1 + dx
            grid = torch.stack([px / (W - 1) * 2 - 1, py / (H - 1) * 2 - 1], dim=-1)          # align_corners=True: -1 .. 1 = pixel 0 .. size-1
            samp = torch.nn.functional.grid_sample(xg, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
            samp = samp * mask[:, g * K + t].double()[:, None]
            want += torch.einsum("bchw,oc->bohw", samp, w[:, g * cg:(g + 1) * cg, ky, kx].double())
    assert torch.allclose(got.double(), want, atol=2e-4, rtol=1e-4), (got.double() - want).abs().max()
