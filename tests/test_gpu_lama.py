"""LaMa on the MI355X through the C-ABI, against the CPU oracle (oracle/lama.py: the restated reference wrapper --
pinned to the reference's own LamaInpaint by tests/golden/wrappers.npz -- around BigLamaNet, the published generator restated with
torch.fft.rfftn / irfftn, reflect-padded convs and conv_transpose2d; the network's parity with big-lama.pt is unpinned: the blob
is missing).  Bar: uint8 outputs equal up to isolated truncation flips of an fp32 sum taken in another order."""
import numpy as np
import pytest
import torch

from oracle.lama import BigLamaNet, LamaOracle
from oracle.sttn_auto import calculate_psnr
from vsr_amd.synth import make_clip, make_lama_state_dict

pytestmark = pytest.mark.gpu


def _case(seed, B, H, W):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    mask = np.zeros((B, H, W), dtype=np.uint8)
    mask[:, H // 3: H // 3 + max(H // 4, 3), W // 8: W - W // 6] = 255
    mask[0, -5:, -9:] = 255
    return img, mask


def _bar(got, ref, what, frac=2e-3):
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    print(f"{what}: max|d| {d.max()}, differing {float((d > 0).mean()):.2e}, PSNR {calculate_psnr(got, ref):.1f} dB")
    assert got.shape == ref.shape and d.max() <= 1 and (d > 0).mean() <= frac, what


@pytest.mark.parametrize("B,H,W,blocks", [(1, 32, 48, 1), (2, 61, 90, 2), (3, 159, 852, 2)])
def test_lama_engine_vs_oracle(built_lib, gpu_device, B, H, W, blocks):
    """vsr_lama_inpaint: padding to x8 (61 -> 64, 90 -> 96, 159 -> 160, 852 -> 856: an odd spectrum length 107), stem im2col, reflect
    halos, FFC blocks with DFT-matrix FourierUnits, transposed-conv phases, sigmoid / blend / u8"""
    from vsr_amd.engine import LamaEngine

    sd = make_lama_state_dict(3, blocks)
    eng = LamaEngine(sd, device=0)
    img, mask = _case(11 + B, B, H, W)
    got = eng.inpaint(torch.from_numpy(img).to(gpu_device), torch.from_numpy(mask).to(gpu_device))
    torch.cuda.synchronize()
    again = eng.inpaint(torch.from_numpy(img).to(gpu_device), torch.from_numpy(mask).to(gpu_device))
    assert torch.equal(got, again), "deterministic"
    ora = LamaOracle(BigLamaNet(sd, blocks))
    ref = np.stack(ora._inpaint_batch([img[i] for i in range(B)], [mask[i][:, :, None] for i in range(B)]) if B > 1
                   else [ora.inpaint(img[0], mask[0][:, :, None])])
    _bar(got.cpu().numpy(), ref, f"LaMa {B}x{H}x{W}, {blocks} blocks")
    hole = mask > 0
    assert (got.cpu().numpy()[hole] != img[hole]).mean() > 0.9
    eng.close()


def test_lama_fourier_unit_vs_torch_fft(built_lib, gpu_device):
    """The FFC block's spectral branch on the GPU (four DFT-matrix GEMMs + the 1x1 conv on stacked re / im) against
    torch.fft.rfftn / irfftn(norm='ortho') on the same input, read back from the workspace (h = 5, w = 14: wf = 8)."""
    from vsr_amd.engine import LamaEngine

    sd = make_lama_state_dict(5, 1)
    eng = LamaEngine(sd, device=0)
    img, mask = _case(2, 1, 40, 112)
    eng.inpaint(torch.from_numpy(img).to(gpu_device), torch.from_numpy(mask).to(gpu_device))
    torch.cuda.synchronize()
    h, w, cs = 5, 14, 192
    s1 = torch.from_numpy(eng.read_buffer(12, h * w * cs).reshape(1, h, w, cs)).permute(0, 3, 1, 2)      # LB_S1 of the last FFC
    s2 = eng.read_buffer(13, h * w * cs).reshape(1, h, w, cs)                                             # LB_S2 = S1 + fu(S1)
    with torch.no_grad():
        fu = BigLamaNet(sd, 1)._fourier_unit(s1, "model.5.conv2.ffc.convg2g.fu")
    ref = (s1 + fu).permute(0, 2, 3, 1).numpy()
    err = float(np.abs(s2 - ref).max())
    print(f"FourierUnit on the GPU vs torch.fft: max abs err {err:.2e} (values up to {np.abs(ref).max():.2f})")
    assert err <= 3e-5 * max(1.0, float(np.abs(ref).max()))
    eng.close()


def test_lama_plugin_vs_oracle(built_lib, gpu_device):
    """LamaInpaint.__call__ / .inpaint / ._inpaint_batch (lama_inpaint.py:17-114): 9 frames -> mini-batches 4 + 4 + 1, strip
    330x61 padded to 336x64, whole strip overwritten (pixels outside the hole may move by one level: x / 255 * 255 in fp32)."""
    from oracle.make_golden_wrappers import LAMA_CLIP
    from vsr_amd.backend.inpaint.lama_inpaint import LamaInpaint
    from vsr_amd.backend.tools.inpaint_tools import create_mask

    c = LAMA_CLIP
    clip = make_clip(c["n"], c["H"], c["W"], c["box"], seed=c["seed"])
    b = c["box"]
    mask = create_mask((c["H"], c["W"]), [(b[2], b[3], b[0], b[1])])
    sd = make_lama_state_dict(7, 2)
    plug = LamaInpaint("cuda:0", sd)
    frames_in = [f.copy() for f in clip]
    got = np.stack(plug(frames_in, mask))
    assert all(np.array_equal(a, b2) for a, b2 in zip(frames_in, clip)), "inputs are not mutated"
    ora = LamaOracle(BigLamaNet(sd, 2))
    ref = np.stack(ora([f for f in clip], mask))
    y0, y1 = 121, 182
    assert np.array_equal(got[:, :y0], clip[:, :y0]), "rows outside the strip are untouched"
    _bar(got[:, y0:y1], ref[:, y0:y1], "LamaInpaint.__call__")
    _bar(plug.inpaint(clip[0], mask), ora.inpaint(clip[0], mask), "LamaInpaint.inpaint (whole frame)")
    _bar(plug._inpaint_batch([clip[1][y0:y1]], [mask[y0:y1, :, None]])[0], ora._inpaint_batch([clip[1][y0:y1]], [mask[y0:y1, :, None]])[0],
         "_inpaint_batch of one image")
    plug.close()


def test_lama_plugin_overlapping_strips(built_lib, gpu_device):
    """two subtitle groups closer than split_h give strips that share rows.  The reference crops every strip from the
    original frames and writes them back afterwards, in order (lama_inpaint.py:88-106); in place, the second strip saw the first
    one's output."""
    from vsr_amd.backend.inpaint.lama_inpaint import LamaInpaint
    from vsr_amd.backend.tools.inpaint_tools import create_mask, get_inpaint_area_by_mask

    H, W = 240, 330
    clip = make_clip(5, H, W, (60, 70, 40, 290), seed=21)
    mask = create_mask((H, W), [(40, 290, 62, 68), (60, 270, 112, 118)])
    areas = get_inpaint_area_by_mask(W, H, int(W * 3 / 16), mask[:, :, None])
    assert len(areas) == 2 and areas[0][1] > areas[1][0], areas           # the strips overlap
    sd = make_lama_state_dict(7, 2)
    plug = LamaInpaint("cuda:0", sd)
    got = np.stack(plug([f.copy() for f in clip], mask))
    ref = np.stack(LamaOracle(BigLamaNet(sd, 2))([f for f in clip], mask))
    _bar(got, ref, "LamaInpaint.__call__ with overlapping strips")
    plug.close()


def test_lama_strip_size_full_network(built_lib, gpu_device):
    """BASELINE config 1's arithmetic at the 1080p strip: all 18 FFC residual blocks (51 M parameters), 1920x360, 2 frames
    (45 x 240 feature maps: DFT lengths 45 = 3^2 5 and 240 = 2^4 3 5)"""
    from vsr_amd.engine import LamaEngine

    sd = make_lama_state_dict(0, 18)
    eng = LamaEngine(sd, device=0)
    assert eng.n_blocks == 18 and abs(eng.flops(1, 360, 1920) / 1e12 - 1.26) < 0.02
    clip = make_clip(2, 360, 1920, (230, 350, 288, 1632), seed=4)
    mask = np.zeros((360, 1920), np.uint8)
    mask[220:360, 278:1642] = 255
    got = eng.inpaint(torch.from_numpy(clip).to(gpu_device), torch.from_numpy(mask).to(gpu_device)).cpu().numpy()
    ora = LamaOracle(BigLamaNet(sd, 18))
    ref = np.stack(ora._inpaint_batch([clip[0], clip[1]], [mask[:, :, None]] * 2))
    _bar(got, ref, "LaMa 18 blocks, 2 x 1920x360", frac=5e-3)
    eng.close()


def test_config1_lama_mode_through_run(built_lib, gpu_device):
    """BASELINE.json configs[0] plumbing: SubtitleRemover.run() with --inpaint-mode lama on an 852x480 clip whose subtitle sits in
    test/test.png's box (y 373..452, x 111..766), detector injected; the same driver with the CPU oracle plugged in is the reference."""
    from vsr_amd.backend.config import config
    from vsr_amd.backend.inpaint.lama_inpaint import LamaInpaint
    from vsr_amd.backend.main import SubtitleRemover
    from vsr_amd.backend.tools.constant import InpaintMode
    from vsr_amd.backend.tools.video_io import ArrayVideo

    H, W, n = 480, 852, 26
    box = (373, 452, 111, 766)
    clip = make_clip(n, H, W, box, seed=12)
    quad = np.array([[[box[2], box[0]], [box[3], box[0]], [box[3], box[1]], [box[2], box[1]]]])

    class Det:
        def predict(self, img):
            return [{"dt_polys": quad}]

    sd = make_lama_state_dict(9, 2)
    old = config.inpaintMode.value
    config.inpaintMode.value = InpaintMode.LAMA
    outs = []
    try:
        for plugin in (LamaInpaint("cuda:0", sd), LamaOracle(BigLamaNet(sd, 2))):
            sr = SubtitleRemover(ArrayVideo(clip.copy(), fps=25.0), device="cuda:0")
            sr.sub_areas = [(0, H, 0, W)]
            sr.text_detector = Det()
            sr.lama_inpaint = plugin
            sr.run()
            outs.append(np.stack(sr.video_writer.frames))
            if hasattr(plugin, "close"):
                plugin.close()
    finally:
        config.inpaintMode.value = old
    got, ref = outs
    assert got.shape == clip.shape
    _bar(got, ref, "SubtitleRemover.run() --inpaint-mode lama")
    assert (got != clip).any(axis=(1, 2, 3)).all(), "every frame carries the subtitle and is repainted"
