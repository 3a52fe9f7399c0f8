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
                                         ("ppocr_det_graph.json", 96, 160), ("ppocr_det_graph.json", 224, 352), ("ppocr_det_graph.json", 544, 960)])
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
    # sigmoid output in [0, 1], both programs at the 1080p net input (544 x 960) too.  Round 3: the synthetic weights are calibrated
    # (oracle/ppocr_det.synthetic_weights: unit-variance activations through all ~150 layers of the server program), so the map is
    # no longer saturated and an ABSOLUTE bar holds: 1e-4 against the fp64 interpreter (measured 1e-6 .. 1e-5; a wrong layer moves
    # the map by 1e-2 or more)
    spread = ((ref64 > 0.05) & (ref64 < 0.95)).double().mean().item()
    assert got.shape == ref64.shape and err <= 1e-4 and spread > 0.5, (err, spread)


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
    assert ea <= 1e-4 and eb <= 1e-4                                   # absolute, on calibrated weights (round 3)
    # round 5: the server program's 64 -> 64 2x2 transposed conv runs as a GEMM too (bias on the GEMM, batch_norm + ReLU on the
    # pass that writes its result); the mobile program's 24-channel one stays on the direct kernel
    n_deconv = sum(1 for k in r._gemm if k[0] == "deconv")
    assert n_deconv == (1 if fixture == "ppocr_det_graph.json" else 0)
    if n_deconv:
        r.use_gemm, r.deconv_gemm = True, False
        c = r.run(x).clone()
        torch.cuda.synchronize()
        ec = (c.cpu().double() - ref64).abs().max().item()
        print(f"{fixture}: transposed conv on the direct kernel instead: err vs fp64 {ec:.2e}; GEMM form vs direct {(a - c).abs().max().item():.2e}")
        assert ec <= 1e-4
    r.close()


@pytest.mark.skipif(os.environ.get("VSR_DET_GRAPH") != "1", reason="HIP-graph replay of the detector is opt-in (round 1's fault: captured memset nodes, fixed in round 6 -- "
                                                                  "profiles/r06_det_graph_triage.log; green 3/3 with VSR_DET_GRAPH=1); set VSR_DET_GRAPH=1 to exercise it")
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
    r.nhwc = "0"                                          # the recorded WALK is under test (the server program's default is the NHWC plan)
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
    r.nhwc = "0"                                          # walk + recorded walk here; the NHWC plan's batches: test_nhwc_plan_batches_and_replays
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


@pytest.mark.parametrize("fixture,nb,H,W", [("ppocr_det_graph.json", 1, 96, 160), ("ppocr_det_graph.json", 2, 224, 352), ("ppocr_det_graph.json", 1, 544, 960),
                                            ("ppocr_det_graph.json", 8, 544, 960), ("ppocr_det_fast_graph.json", 2, 160, 224)])
def test_nhwc_plan_matches_interpreter(built_lib, gpu_device, fixture, nb, H, W):
    """the compiled NHWC-resident plan (ocr_det_nhwc.py; the server program's default path) against the fp64 interpreter -- the bar of
    test_program_matches_interpreter -- and against the op-by-op walk on the NCHW kernels (same fp32 products, batch_norm folded into the
    weights instead of applied after: two fp32 evaluations, each within the bar of the fp64 result, so within twice the bar of each other);
    the mobile program is forced onto a plan (its blocks keep most values NCHW)"""
    g = load_graph(os.path.join(GOLD, fixture))
    w = synthetic_weights(g)
    x = torch.from_numpy(np.random.default_rng(H + W + nb).standard_normal((nb, 3, H, W)).astype(np.float32))
    r = ocr_det.PaddleGraphRunner(g, w, device=0)
    r.nhwc = "1"
    xd = x.to(gpu_device).contiguous()
    got = r.run_planned(xd).clone()
    walk = r.run(xd).clone()
    torch.cuda.synchronize()
    kinds = r.plan_for(xd.shape)["kinds"]
    vs_walk = (got - walk).abs().max().item()
    nref = min(nb, 2)                                       # (the interpreter takes minutes per 1080p frame in fp64)
    ref64 = run_graph(g, w, x[:nref], dtype=torch.float64)
    err = (got[:nref].cpu().double() - ref64).abs().max().item()
    print(f"{fixture} {nb}x{H}x{W}: NHWC plan vs fp64 interpreter {err:.2e}, vs the NCHW walk {vs_walk:.2e}; steps {kinds}")
    assert got.shape == walk.shape and err <= 1e-4 and vs_walk <= 2e-4
    if nb > 2:                                              # the images the interpreter did not see: against the walk
        assert (got[nref:] - walk[nref:]).abs().max().item() <= 2e-4
    if fixture == "ppocr_det_graph.json":
        assert kinds.get("from_view", 0) == 0 and kinds["to_view"] + kinds.get("im2col_view", 0) == 3
        auto = ocr_det.PaddleGraphRunner(g, w, device=0)
        assert auto.nhwc == "auto" and auto.plan_for(xd.shape) is not None          # the server program's default
        auto.close()
    r.close()


def test_nhwc_plan_batches_and_replays(built_lib, gpu_device):
    """every image of a batch comes out of the plan exactly as it does alone (a GEMM row's sum does not depend on the other rows), replays
    on fresh inputs and after another shape are bit-equal to first runs, two instances agree bit for bit"""
    g = load_graph(os.path.join(GOLD, "ppocr_det_graph.json"))
    w = synthetic_weights(g)
    r, r2 = ocr_det.PaddleGraphRunner(g, w, device=0), ocr_det.PaddleGraphRunner(g, w, device=0)
    rng = np.random.default_rng(31)
    x = torch.from_numpy(rng.standard_normal((3, 3, 96, 160)).astype(np.float32)).to(gpu_device)
    y = torch.from_numpy(rng.standard_normal((3, 3, 96, 160)).astype(np.float32)).to(gpu_device)
    single = torch.cat([r.run_taped(x[b:b + 1].contiguous()).clone() for b in range(3)])
    batched = r.run_taped(x).clone()
    other = r.run_taped(y).clone()
    again = r.run_taped(x).clone()
    second = r2.run_taped(x).clone()
    torch.cuda.synchronize()
    assert r.plan_for(x.shape) is not None and not r._tapes
    assert torch.equal(batched, single) and torch.equal(again, batched) and torch.equal(second, batched) and not torch.equal(other, batched)
    r.close()
    r2.close()


def test_detector_lanes_on_the_device(built_lib, gpu_device, monkeypatch):
    """SubtitleDetect's resident pass with two REAL detectors (TextDetection + clone(): own recorded launch lists, intermediates and
    post-process buffers, own stream, own host thread -- the default since round 4) gives the {frame_no: boxes} of one lane, and the
    probability maps of the two instances are bit-equal on the same frames."""
    from vsr_amd.backend.tools.subtitle_detect import SubtitleDetect

    g = load_graph(os.path.join(GOLD, "ppocr_det_fast_graph.json"))
    det = ocr_det.TextDetection(g, synthetic_weights(g), device=0)
    rng = np.random.default_rng(29)
    N, H, W = 41, 270, 480
    frames = torch.from_numpy(rng.integers(0, 256, size=(N, H, W, 3), dtype=np.uint8)).to(gpu_device)

    class Clip:
        def __len__(self):
            return N

    clip = Clip()
    clip.frames = frames
    other = det.clone()
    a, b = det.probability_maps_device(frames[:5]).clone(), other.probability_maps_device(frames[:5]).clone()
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # the same batches over one and over two instances running side by side (own streams, own host threads): bit-equal maps
    from vsr_amd.backend.tools import batch_lanes

    parts = [list(range(s, min(s + 8, N))) for s in range(0, N, 8)]
    fwd = lambda d, part: d.probability_maps_device(frames[torch.tensor(part, device=gpu_device)]).clone()
    one = batch_lanes.run_map(parts, [det], fwd, frames.device)
    two = batch_lanes.run_map(parts, [det, other], fwd, frames.device)
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(one, two))
    got = {}
    for lanes in ("1", "2", "3"):
        monkeypatch.setenv("VSR_DET_LANES", lanes)
        sd = SubtitleDetect(None, [(0, H, 0, W)], text_detector=det)
        sd.SAMPLE_STEP = 1
        got[lanes] = sd._find_resident(None, clip)
    assert got["1"] == got["2"] == got["3"]


from oracle import db_postprocess as dbo
from test_db_postprocess import blob_map


@pytest.mark.parametrize("seed,H,W,nboxes,holes,specks,tilt", [(1, 544, 960, 6, 0, 0, 0.3), (2, 544, 960, 60, 0, 12, 0.3), (3, 96, 160, 3, 0, 4, 0.3),
                                                               (4, 544, 960, 0, 0, 0, 0.3), (5, 544, 960, 10, 0, 30, 1.2), (6, 544, 960, 8, 0, 0, 0.0),
                                                               (7, 272, 480, 8, 0, 40, 0.6), (8, 544, 960, 12, 6, 5, 0.3), (9, 544, 960, 9, 0, 6, 0.2),
                                                               (10, 544, 960, 7, 0, 3, 0.8), (11, 544, 960, 5, 0, 0, 0.05), (12, 1088, 1920, 8, 0, 4, 0.3)])
def test_device_db_postprocess_equals_the_oracle(built_lib, gpu_device, seed, H, W, nboxes, holes, specks, tilt):
    """DBPostProcess on the GPU (threshold, union-find labelling, hull / minimum-area rectangle / fillPoly-rule score / Clipper-rule
    offset / second rectangle per component) against oracle/db_postprocess.py, the restatement of PaddleX's DBPostProcess over
    restated cv2 / pyclipper primitives: rotated, touching, tiny (specks: below min_size) and axis-aligned boxes.
    Bar: the same boxes in the same order, corners EQUAL (both sides compute the geometry in fp64 without contraction, through the
    same float32 / integer truncation points; a corner may still differ by one source pixel where libm's sin / cos / acos / atan2 of the two
    machines disagree in the last bit at a rounding boundary -- allowed on at most 1 % of the coordinates), scores to 1e-6.
    A map with holes (seed 8) takes the host path, which must give the oracle's boxes exactly."""
    prob = blob_map(seed, H, W, nboxes, holes, specks, max_tilt=tilt)
    want_b, want_s = dbo.db_postprocess(prob, 1080, 1920)
    post = ocr_det.DeviceDBPostProcess(gpu_device)
    got_b, got_s = post(torch.from_numpy(prob).to(gpu_device), 1080, 1920)
    assert got_b.shape == want_b.shape, (got_b.shape, want_b.shape)
    if len(want_s):
        diff = np.abs(got_b.astype(np.int64) - want_b.astype(np.int64))
        assert diff.max() <= 1 and (diff != 0).mean() <= 0.01, (diff.max(), (diff != 0).mean())
    assert np.allclose(got_s, want_s, rtol=0, atol=1e-6)
    # which path ran: the device keeps hole-free maps whose components are at most 256 rows tall
    import scipy.ndimage

    ref, n = scipy.ndimage.label(prob > 0.3, structure=np.ones((3, 3), dtype=int))
    n_holes = scipy.ndimage.label(np.pad(~(prob > 0.3), 1, constant_values=True))[1] - 1
    tall = any(sl[0].stop - sl[0].start > 256 for sl in scipy.ndimage.find_objects(ref))
    to_host = n_holes > 0 or tall or n > post.cap
    assert post.host_fallbacks == (1 if to_host else 0), (n_holes, tall, n)
    assert (holes > 0) <= (n_holes > 0)
    print(f"seed {seed}: {n} components, {n_holes} holes, {len(want_s)} boxes, {'host' if to_host else 'device'} path")
    if nboxes >= 6 and tilt <= 0.3:
        assert len(want_s) >= 3
    if to_host:
        assert np.array_equal(got_b, want_b)
    # the labelling itself: same partition as scipy's, labels = raster index of the first pixel
    labels = post._work[(H, W, 0)][0].view(H, W).cpu().numpy()
    assert ((labels >= 0) == (ref > 0)).all()
    flat_ref, flat_lab = ref.reshape(-1), labels.reshape(-1)
    idx = np.nonzero(flat_ref)[0]
    firsts = np.full(n + 1, -1, np.int64)
    for i in idx[::-1]:
        firsts[flat_ref[i]] = i
    assert np.array_equal(flat_lab[idx], firsts[flat_ref[idx]])


@pytest.mark.parametrize("H,W,density", [(17, 37, 0.5), (64, 65, 0.6), (33, 100, 0.45), (8, 64, 0.9), (5, 1, 1.0), (1, 130, 0.7), (40, 63, 0.55), (96, 160, 0.35),
                                         (30, 128, 1.0), (50, 200, 0.62)])
def test_ccl_labels_random_maps(built_lib, gpu_device, H, W, density):
    """vsr_det_launch_ccl (run-start labels by wave ballot, only the unions no neighbour pair implies, one set of statistics atomics per wave)
    on random maps whose rows straddle the 64-pixel waves in every way: the partition is scipy's 8-connected one, every label is the raster
    index of its component's first pixel, area and bounding box per component are exact"""
    import ctypes as C

    import scipy.ndimage

    from vsr_amd._lib import check, lib

    rng = np.random.default_rng(H * 1000 + W)
    prob = (rng.random((H, W)) < density).astype(np.float32) * 0.8 + 0.1
    if H >= 30 and W >= 100:
        prob[5:25, 10:90] = 0.9                                # a solid block: long runs, wave-uniform roots
    ref, n = scipy.ndimage.label(prob > 0.3, structure=np.ones((3, 3), dtype=int))
    cap = max(n, 1) + 4
    i32 = torch.int32
    d = torch.from_numpy(prob).to(gpu_device)
    labels, stats = torch.empty(H * W, dtype=i32, device=gpu_device), torch.empty(H * W * 5, dtype=i32, device=gpu_device)
    comps, count = torch.empty(cap * 6, dtype=i32, device=gpu_device), torch.zeros(4, dtype=i32, device=gpu_device)
    p = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(3):                                         # (the unions race: every run must give the same partition)
        check(lib.vsr_det_launch_ccl(p(d), H, W, C.c_float(0.3), p(labels), p(stats), p(comps), cap, p(count), None))
        torch.cuda.synchronize()
        lab = labels.cpu().numpy().reshape(H, W)
        assert int(count[0].item()) == n
        assert ((lab >= 0) == (ref > 0)).all()
        got = comps.cpu().numpy().reshape(cap, 6)[:n]
        got = got[np.argsort(got[:, 0])]
        want = []
        for k, sl in enumerate(scipy.ndimage.find_objects(ref), start=1):
            ys, xs = np.nonzero(ref == k)
            first = int((ys * W + xs).min())
            assert (lab[ref == k] == first).all()
            want.append((first, len(ys), xs.min(), xs.max(), ys.min(), ys.max()))
        want = np.array(sorted(want), dtype=np.int64).reshape(-1, 6)
        assert np.array_equal(got, want)


def test_device_db_postprocess_batch_equals_per_map(built_lib, gpu_device):
    """DeviceDBPostProcess.batch (every map of a forward in work buffers of its own, ONE read-back -- what predict_batch /
    predict_batch_device use) gives, map by map, exactly what the per-map call gives: boxes, scores, and the host fallback of the map with
    holes; an empty map among them stays empty; a second batch reuses the slots"""
    cases = [(1, 6, 0, 0, 0.3), (8, 12, 6, 5, 0.3), (4, 0, 0, 0, 0.3), (5, 10, 0, 30, 1.2), (9, 9, 0, 6, 0.2), (2, 60, 0, 12, 0.3), (6, 8, 0, 0, 0.0), (11, 5, 0, 0, 0.05)]
    maps = np.stack([blob_map(seed, 544, 960, nb, holes, specks, max_tilt=tilt) for seed, nb, holes, specks, tilt in cases])
    dev = torch.from_numpy(maps).to(gpu_device)
    single, batched = ocr_det.DeviceDBPostProcess(gpu_device), ocr_det.DeviceDBPostProcess(gpu_device)
    want = [single(dev[b], 1080, 1920) for b in range(len(cases))]
    for _ in range(2):
        got = batched.batch(dev, 1080, 1920)
        assert len(got) == len(want)
        for (gb, gs), (wb, ws) in zip(got, want):
            assert np.array_equal(gb, wb) and gs == ws
    assert 1 <= single.host_fallbacks < len(cases) and batched.host_fallbacks == 2 * single.host_fallbacks and len(want[2][1]) == 0 and len(want[0][1]) >= 3
    assert batched.batch(dev[:0], 1080, 1920) == []


def test_device_hole_count(built_lib, gpu_device):
    """the Euler-number hole count that routes a map to the host: enclosed background regions (4-connected), not notches"""
    prob = np.full((64, 96), 0.1, np.float32)
    prob[10:30, 10:50] = 0.9
    prob[15:18, 20:24] = 0.1                                     # one hole
    prob[22, 30] = 0.1                                           # another
    prob[10:14, 40:44] = 0.1                                     # a notch open to the outside: not a hole
    prob[40:50, 20:60] = 0.8
    prob[44, 20:25] = 0.1                                        # a slit from the edge: not a hole
    prob[5, 5] = 0.9
    prob[6, 6] = 0.9                                             # diagonal pair: one component
    post = ocr_det.DeviceDBPostProcess(gpu_device)
    post(torch.from_numpy(prob).to(gpu_device), 64, 96)
    host = post._work[(64, 96, 0)][5].cpu().numpy()
    assert host[0] == 3 and host[1] == 2
    assert post.host_fallbacks == 1


def test_device_db_postprocess_overflow_falls_back(built_lib, gpu_device):
    """more components than the device record list holds, or a component taller than 256 rows: the host statement runs on the
    downloaded map (exactly the oracle's result)"""
    rng = np.random.default_rng(9)
    prob = np.full((256, 384), 0.05, np.float32)
    for y in range(4, 250, 12):                                  # 21 x 31 = 651 isolated 6x7 blobs
        for x in range(4, 376, 12):
            prob[y:y + 6, x:x + 7] = rng.uniform(0.5, 0.95)
    want_b, want_s = dbo.db_postprocess(prob, 512, 768)
    assert len(want_s) > 300
    post = ocr_det.DeviceDBPostProcess(gpu_device, cap=64)
    got_b, got_s = post(torch.from_numpy(prob).to(gpu_device), 512, 768)
    assert post.host_fallbacks == 1 and np.array_equal(got_b, want_b) and np.allclose(got_s, want_s, atol=1e-6)
    tall = np.full((544, 200), 0.05, np.float32)
    tall[20:400, 50:90] = 0.9                                    # 380 rows
    want_b, want_s = dbo.db_postprocess(tall, 1088, 400)
    post = ocr_det.DeviceDBPostProcess(gpu_device)
    got_b, got_s = post(torch.from_numpy(tall).to(gpu_device), 1088, 400)
    assert post.host_fallbacks == 1 and np.array_equal(got_b, want_b) and len(want_s) == 1
