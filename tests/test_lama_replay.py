"""LaMa without a GPU: the engine's plan -- packed weights (BatchNorm folded, FFC branches fused, FourierUnit
channel permutation, transposed-conv phases), offset tables, DFT matrices, op order -- replayed on the CPU against the oracle
network (oracle/lama.py BigLamaNet: torch.fft.rfftn / irfftn, F.conv2d(padding_mode reflect), F.conv_transpose2d)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import _replay_lama  # noqa: E402
from oracle.lama import BigLamaNet, LamaOracle  # noqa: E402
from vsr_amd.synth import lama_state_dict_spec, make_lama_state_dict  # noqa: E402


def _case(seed, B, H, W):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    mask = np.zeros((B, H, W), dtype=np.uint8)
    mask[:, H // 3: H // 3 + max(H // 4, 3), W // 8: W - W // 6] = 255
    mask[0, -5:, -9:] = 255                                   # touches the bottom-right corner: the symmetric padding of the mask matters
    return img, mask


@pytest.mark.parametrize("B,H,W,blocks", [(1, 32, 48, 1), (2, 61, 90, 2)])
def test_lama_plan_replay_matches_oracle(built_lib, B, H, W, blocks):
    from vsr_amd.engine import LamaEngine

    sd = make_lama_state_dict(3, blocks)
    eng = LamaEngine(sd, device=None)
    assert eng.n_blocks == blocks
    view = _replay_lama.lama_plan_view(built_lib, eng, B, H, W)
    img, mask = _case(11 + B, B, H, W)
    got, bufs = _replay_lama.replay_lama(view, eng.packed_weights(), img, mask)
    ora = LamaOracle(BigLamaNet(sd, blocks))
    ref = np.stack(ora._inpaint_batch([img[i] for i in range(B)], [mask[i][:, :, None] for i in range(B)]) if B > 1
                   else [ora.inpaint(img[0], mask[0][:, :, None])])
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    hole = mask > 0
    print(f"LaMa replay {B}x{H}x{W}, {blocks} blocks: max|d| {d.max()}, differing {float((d > 0).mean()):.2e}; "
          f"hole pixels changed: {float((got[hole] != img[hole]).mean()):.3f}")
    assert got.shape == ref.shape
    assert d.max() <= 1 and (d > 0).mean() < 2e-3            # fp32 sums in another order: isolated u8 truncation flips
    assert (got[hole] != img[hole]).mean() > 0.9, "the hole is repainted"
    # algorithmic FLOPs: every conv / DFT stage counted once
    assert view.flops == pytest.approx(eng.flops(B, H, W))
    view.close()
    eng.close()


def test_lama_fourier_unit_stage_by_stage(built_lib):
    """the four DFT GEMMs of one FourierUnit against torch.fft on the same S1 tensor (odd spectrum width w/2+1, h != w)"""
    from vsr_amd.engine import LamaEngine

    sd = make_lama_state_dict(5, 1)
    eng = LamaEngine(sd, device=None)
    B, H, W = 1, 40, 112                                      # h = 5, w = 14, wf = 8
    view = _replay_lama.lama_plan_view(built_lib, eng, B, H, W)
    img, mask = _case(2, B, H, W)
    _, bufs = _replay_lama.replay_lama(view, eng.packed_weights(), img, mask)
    h, w, cs = 5, 14, 192
    # the LAST FourierUnit's input (S1) and output (S2 = S1 + fu(S1)) are still in the buffers
    s1 = torch.from_numpy(bufs[12][: h * w * cs].reshape(1, h, w, cs).copy()).permute(0, 3, 1, 2)
    s2 = bufs[13][: h * w * cs].reshape(1, h, w, cs)
    net = BigLamaNet(sd, 1)
    with torch.no_grad():
        fu = net._fourier_unit(s1, "model.5.conv2.ffc.convg2g.fu")
    ref = (s1 + fu).permute(0, 2, 3, 1).numpy()
    err = np.abs(s2 - ref).max()
    print(f"FourierUnit via DFT matrices vs torch.fft: max abs err {err:.2e} (values up to {np.abs(ref).max():.2f})")
    assert err <= 2e-5 * max(1.0, float(np.abs(ref).max()))
    view.close()
    eng.close()


def test_lama_state_dict_contract(built_lib):
    """big-lama's generator: 51 M parameters, strict key handling (unknown prefixes rejected, `generator.` prefix and
    num_batches_tracked accepted, a missing tensor fails finalize)"""
    from vsr_amd import _lib
    from vsr_amd.engine import LamaEngine

    spec = lama_state_dict_spec(18)
    n_params = sum(int(np.prod(s)) for k, s in spec if not k.endswith(("running_mean", "running_var")))
    assert 50_500_000 < n_params < 51_500_000
    sd = make_lama_state_dict(0, 1)
    pref = {"generator." + k: v for k, v in sd.items()}
    pref["generator.model.1.bn_l.num_batches_tracked"] = np.zeros((), np.float32)
    a, b = LamaEngine(sd, device=None), LamaEngine(pref, device=None)
    assert np.array_equal(a.packed_weights(), b.packed_weights())
    a.close()
    b.close()
    with pytest.raises(_lib.VsrError):
        LamaEngine({**sd, "evaluator.foo": np.zeros(3, np.float32)}, device=None)
    bad = dict(sd)
    del bad["model.5.conv1.ffc.convg2g.fu.bn.running_var"]
    with pytest.raises(_lib.VsrError):
        LamaEngine(bad, device=None)
