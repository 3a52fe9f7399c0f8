"""Text detector on the MI355X: the program executed by csrc/det_kernels.hip against the CPU interpreter
(oracle/ppocr_det.py; parity with Paddle's binary is unpinned -- no Paddle, no weights in the reference mount)."""
import os

import numpy as np
import pytest
import torch

from oracle import cv2_restate as cv2r
from oracle.ppocr_det import run_graph, synthetic_weights
from vsr_amd.backend.tools import ocr_det
from vsr_amd.backend.tools.paddle_graph import load_graph

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("fixture,H,W", [("ppocr_det_fast_graph.json", 96, 160), ("ppocr_det_fast_graph.json", 544, 960),
                                         ("ppocr_det_graph.json", 96, 160), ("ppocr_det_graph.json", 224, 352)])
def test_program_matches_interpreter(built_lib, gpu_device, fixture, H, W):
    g = load_graph(os.path.join(GOLD, fixture))
    w = synthetic_weights(g)
    x = torch.from_numpy(np.random.default_rng(H + W).standard_normal((1, 3, H, W)).astype(np.float32))
    ref64 = run_graph(g, w, x, dtype=torch.float64)
    ref32 = run_graph(g, w, x).double()
    got = ocr_det.PaddleGraphRunner(g, w, device=0).run(x.to(gpu_device).contiguous())
    torch.cuda.synchronize()
    err = (got.cpu().double() - ref64).abs().max().item()
    cpu32 = (ref32 - ref64).abs().max().item()
    print(f"{fixture} {H}x{W}: max abs err of the probability map vs the fp64 interpreter {err:.2e} (fp32 CPU interpreter: {cpu32:.2e})")
    # sigmoid output in [0, 1].  The 22 M-parameter server program is ~150 layers deep and saturates with synthetic weights, so
    # fp32 rounding alone moves it by 1e-3..1e-2: the bar is the fp32 CPU interpreter's own distance from fp64
    assert got.shape == ref64.shape and err <= 3 * cpu32 + 1e-4


def test_predict_preprocessing_and_plumbing(built_lib, gpu_device):
    """resize_long 960 (cv2 INTER_LINEAR, bit-exact vs the restated cv2) + NormalizeImage; predict() returns paddleocr's dict"""
    g = load_graph(os.path.join(GOLD, "ppocr_det_fast_graph.json"))
    det = ocr_det.TextDetection(g, synthetic_weights(g), device=0)
    img = np.random.default_rng(3).integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    prob, rh, rw = det.probability_map(img)
    assert (rh, rw) == (544, 960) and tuple(prob.shape) == (544, 960)
    small = cv2r.resize_linear(img, (rw, rh))
    x = ((small.astype(np.float32) * np.float32(1.0 / 255.0) - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32))
    ref = run_graph(g, synthetic_weights(g), torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))[None])
    assert (prob.cpu() - ref[0, 0]).abs().max().item() <= 5e-4
    res = det.predict(img)
    assert isinstance(res, list) and res[0]["dt_polys"].ndim == 3 and res[0]["dt_polys"].shape[1:] == (4, 2)
    assert len(res[0]["dt_scores"]) == res[0]["dt_polys"].shape[0]


@pytest.mark.parametrize("fixture,H,W", [("ppocr_det_fast_graph.json", 160, 224), ("ppocr_det_graph.json", 160, 224)])
def test_gemm_convs_agree_with_direct_convs(built_lib, gpu_device, fixture, H, W):
    """the dense convolutions on the gather-GEMM (default) against the same program on the direct kernels: the same fp32 sums in
    another order, so both stay within the interpreter's own fp32 rounding of the fp64 result; the folded batch_norm + ReLU
    epilogue must not change what the separate kernels compute"""
    g = load_graph(os.path.join(GOLD, fixture))
    w = synthetic_weights(g)
    x = torch.from_numpy(np.random.default_rng(7).standard_normal((1, 3, H, W)).astype(np.float32)).to(gpu_device)
    r = ocr_det.PaddleGraphRunner(g, w, device=0)
    assert r.use_gemm and any(b is not None for b, _ in r._fuse.values())
    a = r.run(x).clone()
    n_plans = len(r._gemm)
    a2 = r.run(x).clone()                                   # second run reuses the resident plans
    r.use_gemm = False
    b = r.run(x).clone()
    torch.cuda.synchronize()
    ref64 = run_graph(g, w, x.cpu(), dtype=torch.float64)
    cpu32 = (run_graph(g, w, x.cpu()).double() - ref64).abs().max().item()
    ea, eb = (a.cpu().double() - ref64).abs().max().item(), (b.cpu().double() - ref64).abs().max().item()
    print(f"{fixture}: {n_plans} convs on the gather-GEMM; err vs fp64 {ea:.2e} (GEMM) / {eb:.2e} (direct) / {cpu32:.2e} (fp32 CPU)")
    assert n_plans > 10 and len(r._gemm) == n_plans and torch.equal(a, a2)
    assert ea <= 5 * cpu32 + 1e-4 and eb <= 5 * cpu32 + 1e-4          # measured: 3.1x / 2.3x on the saturating server program
    r.close()


@pytest.mark.skipif(os.environ.get("VSR_DET_GRAPH") != "1", reason="HIP-graph replay of the detector is experimental (faulted on its first run); "
                                                                  "set VSR_DET_GRAPH=1 to exercise it")
@pytest.mark.parametrize("fixture", ["ppocr_det_fast_graph.json", "ppocr_det_graph.json"])
def test_graph_replay_equals_eager(built_lib, gpu_device, fixture):
    """the forward replayed from a captured HIP graph (opt-in) is bit-identical to the launch-by-launch pass, for
    new inputs of the captured shape and after switching between shapes"""
    g = load_graph(os.path.join(GOLD, fixture))
    r = ocr_det.PaddleGraphRunner(g, synthetic_weights(g), device=0)
    rng = np.random.default_rng(11)
    xs = [torch.from_numpy(rng.standard_normal((1, 3, h, w)).astype(np.float32)).to(gpu_device) for h, w in ((96, 160), (96, 160), (128, 96), (96, 160))]
    for x in xs:
        eager = r.run(x).clone()
        replay = r.run_graphed(x).clone()
        torch.cuda.synchronize()
        assert torch.equal(eager, replay)
    assert len(r._graphs) == 2
    r.close()


@pytest.mark.parametrize("fixture,H,W", [("ppocr_det_fast_graph.json", 160, 224), ("ppocr_det_graph.json", 544, 960)])
def test_recorded_launch_list_replay(built_lib, gpu_device, fixture, H, W):
    """run_taped: the second pass for an input shape records every launch, later passes replay the list -- bit-identical to the
    op-by-op walk, on fresh inputs, also after another shape was recorded in between (544x960 is the 1080p net input at which
    the HIP-graph replay of round 1 faulted)."""
    g = load_graph(os.path.join(GOLD, fixture))
    r = ocr_det.PaddleGraphRunner(g, synthetic_weights(g), device=0)
    rng = np.random.default_rng(17)
    xs = [torch.from_numpy(rng.standard_normal((1, 3, H, W)).astype(np.float32)).to(gpu_device) for _ in range(3)]
    small = torch.from_numpy(rng.standard_normal((1, 3, 96, 160)).astype(np.float32)).to(gpu_device)
    want = [r.run(x).clone() for x in xs]
    got0 = r.run_taped(xs[0]).clone()                     # records
    r.run_taped(small)                                    # another shape in between
    got1 = r.run_taped(xs[1]).clone()                     # replays
    r.run_taped(small)
    got2 = r.run_taped(xs[2]).clone()
    got0b = r.run_taped(xs[0]).clone()
    torch.cuda.synchronize()
    assert len(r._tapes[tuple(xs[0].shape)][0]) > 100
    for a, b in zip((got0, got1, got2, got0b), (want[0], want[1], want[2], want[0])):
        assert torch.equal(a, b)
    r.close()


@pytest.mark.parametrize("fixture", ["ppocr_det_fast_graph.json", "ppocr_det_graph.json"])
def test_batched_forward_equals_single_frames(built_lib, gpu_device, fixture):
    """predict_batch / a batch through the runner: the frames are independent, so every image of a batch must come out exactly
    as it does alone (batched layout kernels, GEMM row tables over [n][Hp][Wp][Cp], strided channel concat, recorded replay)."""
    g = load_graph(os.path.join(GOLD, fixture))
    r = ocr_det.PaddleGraphRunner(g, synthetic_weights(g), device=0)
    rng = np.random.default_rng(23)
    x = torch.from_numpy(rng.standard_normal((3, 3, 96, 160)).astype(np.float32)).to(gpu_device)
    single = torch.cat([r.run(x[b:b + 1].contiguous()).clone() for b in range(3)])
    batched = r.run(x).clone()
    taped = [r.run_taped(x).clone() for _ in range(2)][-1]
    torch.cuda.synchronize()
    assert batched.shape == single.shape == (3, 1, 96, 160)
    assert torch.equal(batched, single) and torch.equal(taped, single)
    r.close()
    det = ocr_det.TextDetection(g, synthetic_weights(g), device=0)
    imgs = [rng.integers(0, 256, size=(270, 480, 3), dtype=np.uint8) for _ in range(3)]
    maps = det.probability_maps(imgs)
    for b, img in enumerate(imgs):
        assert torch.equal(maps[b], det.probability_map(img)[0])


def _blob_map(seed, H, W, nboxes):
    """probability map with rotated text-like boxes, touching pairs, specks and diagonal (8-connected only) links"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    prob = rng.random((H, W)).astype(np.float32) * 0.25
    for _ in range(nboxes):
        cy, cx = rng.uniform(10, H - 10), rng.uniform(30, W - 30)
        hw, hh, th = rng.uniform(8, 120), rng.uniform(3, 14), rng.uniform(-0.3, 0.3)
        u, v = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th), -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        inside = (np.abs(u) <= hw) & (np.abs(v) <= hh)
        prob[inside] = np.maximum(prob[inside], rng.uniform(0.55, 0.95))
    for k in range(40):                                   # staircases: pixels that touch only diagonally
        y, x = int(rng.integers(2, H - 12)), int(rng.integers(2, W - 12))
        for j in range(8):
            prob[y + j, x + j] = 0.9
    return prob


@pytest.mark.parametrize("seed,H,W,nboxes", [(1, 544, 960, 6), (2, 544, 960, 60), (3, 96, 160, 3), (4, 544, 960, 0)])
def test_device_db_postprocess_equals_host(built_lib, gpu_device, seed, H, W, nboxes):
    """threshold + 8-connected union-find labelling + hull / minimum-area rectangle / score / unclip per component, all on the GPU:
    the boxes and scores of DBPostProcess must equal the all-host version (scipy labelling of the downloaded map, numpy geometry).
    Both sides compute in fp64 but not in the same operation order (BLAS projections on the host), so a corner may land on the
    other side of a rounding boundary: at most one source pixel, scores to 1e-5 (fp32 mean on the host, fp64 on the device)."""
    prob = _blob_map(seed, H, W, nboxes)
    want_b, want_s = ocr_det.db_postprocess(prob, 1080, 1920)
    post = ocr_det.DeviceDBPostProcess(gpu_device)
    got_b, got_s = post(torch.from_numpy(prob).to(gpu_device), 1080, 1920)
    assert got_b.shape == want_b.shape
    if len(want_s):
        assert np.abs(got_b.astype(np.int64) - want_b).max() <= 1, np.abs(got_b.astype(np.int64) - want_b).max()
        assert (got_b != want_b).mean() <= 0.02
    assert np.allclose(got_s, want_s, rtol=0, atol=1e-5)
    if nboxes >= 6:
        assert len(want_s) >= 1
    # the labelling itself: same partition as scipy's, labels = raster index of the first pixel
    import scipy.ndimage

    ref, n = scipy.ndimage.label(prob > 0.3, structure=np.ones((3, 3), dtype=int))
    labels = post._work[(H, W)][0].view(H, W).cpu().numpy()
    assert ((labels >= 0) == (ref > 0)).all()
    first = {}
    flat_ref, flat_lab = ref.reshape(-1), labels.reshape(-1)
    idx = np.nonzero(flat_ref)[0]
    firsts = np.full(n + 1, -1, np.int64)
    for i in idx[::-1]:
        firsts[flat_ref[i]] = i
    assert np.array_equal(flat_lab[idx], firsts[flat_ref[idx]])


def test_device_db_postprocess_overflow_falls_back(built_lib, gpu_device):
    """more components than the device record list holds: the labelling stays on the device, the polygon work runs in numpy on the
    downloaded rows (exactly the host result); a map with more than 4096 components is labelled on the host"""
    rng = np.random.default_rng(9)
    prob = np.full((256, 384), 0.05, np.float32)
    for y in range(4, 250, 12):                                  # 21 x 31 = 651 isolated 6x7 blobs
        for x in range(4, 376, 12):
            prob[y:y + 6, x:x + 7] = rng.uniform(0.5, 0.95)
    want_b, want_s = ocr_det.db_postprocess(prob, 512, 768)
    assert len(want_s) > 300
    got_b, got_s = ocr_det.DeviceDBPostProcess(gpu_device, cap=64)(torch.from_numpy(prob).to(gpu_device), 512, 768)
    assert np.array_equal(got_b, want_b) and np.allclose(got_s, want_s)
    noise = rng.random((256, 384)).astype(np.float32)            # one giant component and a few specks: whatever path, same boxes
    want_b, want_s = ocr_det.db_postprocess(noise, 512, 768)
    got_b, got_s = ocr_det.DeviceDBPostProcess(gpu_device, cap=64)(torch.from_numpy(noise).to(gpu_device), 512, 768)
    assert got_b.shape == want_b.shape and (len(want_s) == 0 or np.abs(got_b.astype(np.int64) - want_b).max() <= 1)
    assert np.allclose(got_s, want_s, atol=1e-5)
