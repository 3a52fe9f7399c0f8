"""Scene-cut pass of propainter mode.  CPU part: the oracle (oracle/scene_cuts.py) against the fixture the
reference's own SceneManager + ContentDetector produced (tests/golden/scene_cuts.json, oracle/make_golden.py) and against
hand-derived known answers of cvtColor(BGR2HSV); the host cut logic.  GPU part: the HIP kernels against the oracle, bit-exact."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from oracle import scene_cuts as sc
from vsr_amd import _lib
from vsr_amd.backend.tools import scene_detect

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def scene_clip(seed=5, n=90, H=120, W=780):
    """the clip generator of oracle/make_golden.py (kept in step with it; the fixture's scores would expose a drift)"""
    rng = np.random.default_rng(seed)
    cuts_at = {18, 25, 33, 50, 51, 70, 88}
    yy, xx = np.mgrid[0:H, 0:W]
    frames, base = [], None
    for i in range(n):
        if base is None or i in cuts_at:
            base = rng.integers(0, 256, (H // 12 + 1, W // 12 + 1, 3)).astype(np.float64)
            shade = rng.uniform(0.3, 1.0)
        img = base[yy // 12, xx // 12] * shade + 6.0 * np.sin(0.3 * i + xx / 40.0)[..., None]
        frames.append(np.clip(img, 0, 255).astype(np.uint8))
    return np.stack(frames)


def _golden():
    with open(os.path.join(GOLD, "scene_cuts.json")) as f:
        return json.load(f)


def test_bgr2hsv_known_answers():
    # (B, G, R) -> (H, S, V): primaries / secondaries at H = 0, 30, 60, 90, 120, 150; greys have H = S = 0;
    # (50,100,150): V=150, S=255*100/150=170, H=60*(50/100)/2=15; (200,40,90): max B: H=(240+60*(90-40)/160)/2=129.4->129, S=204
    cases = {(0, 0, 255): (0, 255, 255), (0, 255, 255): (30, 255, 255), (0, 255, 0): (60, 255, 255), (255, 255, 0): (90, 255, 255),
             (255, 0, 0): (120, 255, 255), (255, 0, 255): (150, 255, 255), (0, 0, 0): (0, 0, 0), (255, 255, 255): (0, 0, 255),
             (77, 77, 77): (0, 0, 77), (50, 100, 150): (15, 170, 150), (200, 40, 90): (129, 204, 200), (10, 0, 255): (179, 255, 255)}
    src = np.array(list(cases), dtype=np.uint8)[None]
    got = sc.bgr2hsv_u8(src)[0]
    for (bgr, want), g in zip(cases.items(), got):
        assert tuple(int(x) for x in g) == want, (bgr, tuple(g), want)
    # H stays inside [0, 180), S and V inside u8, over a dense sample
    rng = np.random.default_rng(0)
    hsv = sc.bgr2hsv_u8(rng.integers(0, 256, (1, 200000, 3), dtype=np.uint8))
    assert hsv[..., 0].max() < 180


def test_oracle_matches_reference_scene_manager():
    for tag, v in _golden().items():
        clip = scene_clip(**v["clip"])
        n, H, W, _ = clip.shape
        w, h, f = sc.downscale_size(W, H)
        assert f == v["factor"]
        scores = [0.0] + sc.scores_from_sums(sc.frame_sums(clip), w * h)
        assert scores == v["scores"], tag                  # same integers, same float expression: exact
        assert sc.scene_div_frame_no(clip) == v["div"], tag
    assert sc.scene_div_frame_no(scene_clip(n=1)) == []


def test_host_cut_logic_follows_the_fixture_scores():
    det = scene_detect.ContentDetector.__new__(scene_detect.ContentDetector)          # the cut logic needs no device
    det.threshold, det.min_scene_len = scene_detect.THRESHOLD, scene_detect.MIN_SCENE_LEN
    for tag, v in _golden().items():
        clip = scene_clip(**v["clip"])
        _, H, W, _ = clip.shape
        w, h, _ = sc.downscale_size(W, H)
        cuts = det.process(sc.frame_sums(clip), w * h)
        assert [c + 1 for c in cuts] == v["div"], tag
    assert scene_detect.compute_downscale_factor(1920) == 7 and scene_detect.compute_downscale_factor(255) == 1
    # exactly at the threshold counts; a cut closer than min_scene_len to the previous one does not
    sums = np.zeros((40, 3), np.int64)
    sums[14] = sums[20] = sums[29] = (2700, 2700, 2700)           # frames 15, 21, 30 with 100 pixels -> score 27.0
    assert det.process(sums, 100) == [15, 30]
    sums[14] = (2700, 2700, 2699)
    assert det.process(sums, 100) == [21]


def test_scene_detection_needs_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        scene_detect.ContentDetector()


def _p(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.gpu
def test_gpu_bgr2hsv_exhaustive(built_lib, gpu_device):
    """every one of the 2^24 BGR triples, bit-exact against the oracle"""
    v = np.arange(1 << 24, dtype=np.uint32)
    src = np.stack([v & 255, (v >> 8) & 255, v >> 16], axis=-1).astype(np.uint8)
    d = torch.from_numpy(src).to(gpu_device)
    out = torch.empty_like(d)
    _lib.check(_lib.lib.vsr_launch_bgr2hsv_u8(_p(d), _p(out), src.shape[0], C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for lo in range(0, 1 << 24, 1 << 22):
        assert np.array_equal(got[lo:lo + (1 << 22)], sc.bgr2hsv_u8(src[lo:lo + (1 << 22)]))


@pytest.mark.gpu
@pytest.mark.parametrize("tag,batch", [("a", 64), ("a", 7), ("b", 64), ("c", 16)])
def test_gpu_frame_sums_and_cuts(built_lib, gpu_device, tag, batch):
    """resize + HSV + |difference| sums: the exact integers of the oracle, for clips with and without down-scaling and for batch
    sizes that carry the previous frame across uploads; the cuts equal the reference's"""
    from vsr_amd.backend.tools.video_io import ArrayVideo

    v = _golden()[tag]
    clip = scene_clip(**v["clip"])
    n, H, W, _ = clip.shape
    det = scene_detect.ContentDetector(device=0, batch_frames=batch)
    sums, npix = det.frame_sums(iter(clip), H, W)
    w, h, _ = sc.downscale_size(W, H)
    assert npix == w * h and sums.shape == (n - 1, 3)
    assert np.array_equal(sums, sc.frame_sums(clip))
    assert scene_detect.get_scene_div_frame_no(ArrayVideo(clip, fps=25.0), detector=det) == v["div"]


@pytest.mark.gpu
def test_gpu_scene_cuts_at_1080p_and_single_frame(built_lib, gpu_device):
    from vsr_amd.backend.tools.video_io import ArrayVideo

    clip = scene_clip(seed=11, n=40, H=1080, W=1920)
    assert scene_detect.get_scene_div_frame_no(ArrayVideo(clip, fps=25.0)) == sc.scene_div_frame_no(clip) == [19, 34]
    assert scene_detect.get_scene_div_frame_no(ArrayVideo(clip[:1], fps=25.0)) == []


@pytest.mark.gpu
def test_gpu_propainter_mode_cuts_intervals_at_detected_scenes(built_lib, gpu_device):
    """SubtitleRemover.propainter_mode without injected scene points (reference main.py:165-167): the interval with text is cut
    where the device pass finds a new scene"""
    from vsr_amd.backend.main import SubtitleRemover
    from vsr_amd.backend.tools.video_io import ArrayVideo

    clip = scene_clip(seed=5, n=60, H=120, W=780).copy()          # scenes start at 18, 33 (25 is too close to 18), 50
    clip[:, 0, 0, 0] = np.arange(60)
    quad = np.array([[[300, 80], [480, 80], [480, 100], [300, 100]]])

    class Det:
        def predict(self, img):
            return [{"dt_polys": quad if 10 <= int(img[0, 0, 0]) < 45 else np.zeros((0, 4, 2))}]

    seen = []

    def plugin(batch, mask):
        seen.append([int(f[0, 0, 0]) for f in batch])
        return [f.copy() for f in batch]

    sr = SubtitleRemover(ArrayVideo(clip, fps=25.0), device="cuda:0", model_path="unused")
    sr.sub_areas = [(0, 120, 0, 780)]
    sr.propainter_mode(None, propainter_inpaint=plugin, text_detector=Det())
    assert sc.scene_div_frame_no(clip) == [19, 34, 51]
    starts = [b[0] for b in seen]
    assert 18 in starts and 33 in starts, seen                      # 0-based first frames of the scenes inside the text interval
    assert all(not (b[0] < 18 <= b[-1]) and not (b[0] < 33 <= b[-1]) for b in seen), seen
    assert len(sr.video_writer.frames) == 60


def test_propainter_mode_asks_for_scene_cuts_when_none_are_given(monkeypatch):
    """host wiring (reference main.py:165-167) without a device: get_scene_div_frame_no stands in through the oracle"""
    from vsr_amd.backend.main import SubtitleRemover
    from vsr_amd.backend.tools.subtitle_detect import SubtitleDetect
    from vsr_amd.backend.tools.video_io import ArrayVideo

    clip = scene_clip(seed=5, n=60, H=120, W=780).copy()
    clip[:, 0, 0, 0] = np.arange(60)
    asked = []

    def fake(v_path, device=0):
        asked.append(device)
        return sc.scene_div_frame_no(v_path.frames)

    monkeypatch.setattr(SubtitleDetect, "get_scene_div_frame_no", staticmethod(fake))
    quad = np.array([[[300, 80], [480, 80], [480, 100], [300, 100]]])

    class Det:
        def predict(self, img):
            return [{"dt_polys": quad if 10 <= int(img[0, 0, 0]) < 45 else np.zeros((0, 4, 2))}]

    seen = []
    sr = SubtitleRemover(ArrayVideo(clip, fps=25.0), device="cuda:3", model_path="unused")
    sr.sub_areas = [(0, 120, 0, 780)]
    sr.propainter_mode(None, propainter_inpaint=lambda b, m: seen.append((int(b[0][0, 0, 0]), int(b[-1][0, 0, 0]))) or [f.copy() for f in b],
                       text_detector=Det())
    assert asked == [3]                                        # the device index of "cuda:3"
    assert seen == [(10, 17), (18, 32), (33, 44)]              # the text interval cut at the scenes starting at frames 18 and 33
    assert len(sr.video_writer.frames) == 60
    seen.clear()
    sr2 = SubtitleRemover(ArrayVideo(clip, fps=25.0), device="cuda:3", model_path="unused")
    sr2.sub_areas = [(0, 120, 0, 780)]
    sr2.propainter_mode(None, propainter_inpaint=lambda b, m: seen.append(len(b)) or [f.copy() for f in b], text_detector=Det(), scene_div_points=[])
    assert asked == [3] and seen == [35]                       # given (even empty) scene points are used as they are


def test_bgr2hsv_restatement_agrees_with_colorsys():
    """opencv is absent, so cvtColor(BGR2HSV) on uint8 stays "parity unpinned" -- but its integer algorithm approximates the
    textbook conversion that Python's own colorsys implements (written by other people): H in half degrees, S and V scaled to 255.
    On a grid over the BGR cube the restatement stays within one level of the rounded textbook values (hue compared on the
    circle; grey pixels have no hue).  Catches a wrong sector, channel order or scale -- not the last bit of OpenCV's rounding."""
    import colorsys

    import numpy as np

    from oracle.scene_cuts import bgr2hsv_u8

    g = np.array(sorted(set(list(range(0, 256, 15)) + [1, 2, 127, 128, 254, 255])), dtype=np.uint8)
    b, gg, r = np.meshgrid(g, g, g, indexing="ij")
    img = np.stack([b.reshape(-1), gg.reshape(-1), r.reshape(-1)], axis=-1)[None]
    got = bgr2hsv_u8(img)[0].astype(int)
    worst_h = worst_s = 0
    for (bb, g_, rr), (h, s, v) in zip(img[0].tolist(), got.tolist()):
        hh, ss, vv = colorsys.rgb_to_hsv(rr / 255.0, g_ / 255.0, bb / 255.0)
        assert v == max(bb, g_, rr) == round(vv * 255)
        worst_s = max(worst_s, abs(s - ss * 255))
        if max(bb, g_, rr) - min(bb, g_, rr) > 0:
            dh = abs(h - hh * 180.0)
            worst_h = max(worst_h, min(dh, 180.0 - dh))
        assert 0 <= h < 180
    assert worst_s <= 1.0 and worst_h <= 1.0, (worst_s, worst_h)
