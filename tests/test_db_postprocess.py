"""DBPostProcess (what TextDetection.predict does after the forward, backend/tools/subtitle_detect.py:41-63,
backend/models/V5/ch_det/inference.yml:49-53).

Part 1 holds oracle/db_postprocess.py -- the restatement of PaddleX's DBPostProcess and of the cv2 / pyclipper primitives under it,
parity unpinned because none of those packages exists here -- to DEFINITIONS its primitives must satisfy whatever the
implementation.  Part 2 holds the product's host path (backend/tools/ocr_det.db_postprocess) to the oracle; the device path is held
to it in tests/test_gpu_ocr_det.py."""
import math

import numpy as np
import pytest
import scipy.ndimage

from oracle import db_postprocess as O


def blob_map(seed, H, W, nboxes, holes=0, specks=0, max_tilt=0.3):
    """probability map with rotated text-like boxes (some touching), optional holes punched into them, specks and staircases"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    prob = rng.random((H, W)).astype(np.float32) * 0.25
    centres = []
    for _ in range(nboxes):
        cy, cx = rng.uniform(10, H - 10), rng.uniform(30, W - 30)
        hw, hh, th = rng.uniform(8, 120), rng.uniform(3, 14), rng.uniform(-max_tilt, max_tilt)
        u, v = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th), -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        inside = (np.abs(u) <= hw) & (np.abs(v) <= hh)
        prob[inside] = np.maximum(prob[inside], rng.uniform(0.55, 0.95))
        centres.append((int(cy), int(cx)))
    for k in range(holes):                                # holes of 1..4 pixels inside boxes
        cy, cx = centres[k % max(len(centres), 1)] if centres else (H // 2, W // 2)
        s = 1 + k % 3
        y, x = min(max(cy, 1), H - 5), min(max(cx + 3 * k, 1), W - 5)
        prob[y:y + s, x:x + s + (k % 2)] = 0.1
    for k in range(specks):
        y, x = int(rng.integers(1, H - 3)), int(rng.integers(1, W - 3))
        prob[y:y + 1 + k % 2, x:x + 1 + (k // 2) % 3] = 0.9
    for k in range(10 if nboxes else 0):                  # staircases: pixels that touch only diagonally
        y, x = int(rng.integers(2, H - 12)), int(rng.integers(2, W - 12))
        for j in range(8):
            prob[y + j, x + j] = 0.9
    return prob


# ---------------------------------------------------------------------------------------------------------------- part 1
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_contours_are_the_borders_of_the_connected_components(seed):
    """one outer border per 8-connected component, one hole border per background region that does not reach the frame
    (4-connected, the dual connectivity); an outer border's points all lie in its component and span its convex hull; every
    point of a border has a background 4-neighbour; borders come in reverse order of their raster-first pixel"""
    prob = blob_map(seed, 96, 160, 5, holes=4, specks=6)
    bm = prob > 0.3
    contours = O.find_contours(bm)
    lab, n = scipy.ndimage.label(bm, structure=np.ones((3, 3), int))
    pad = np.pad(~bm, 1, constant_values=True)
    bl, nb = scipy.ndimage.label(pad)                     # 4-connected background, the frame is background
    n_holes = nb - 1
    assert len(contours) == n + n_holes
    seen_outer = set()
    bm_pad = np.pad(bm, 1)
    for c in contours:
        xs, ys = c[:, 0], c[:, 1]
        assert bm[ys, xs].all()
        bg4 = (~bm_pad[ys + 1, xs]) | (~bm_pad[ys + 1, xs + 2]) | (~bm_pad[ys, xs + 1]) | (~bm_pad[ys + 2, xs + 1])
        assert bg4.all()
        labs = set(lab[ys, xs].tolist())
        assert len(labs) == 1
    # outer borders: exactly the components, hull of the border == hull of the component
    for k in range(1, n + 1):
        yy, xx = np.nonzero(lab == k)
        want = set(O._hull(np.stack([xx, yy], 1)))
        match = [c for c in contours if set(O._hull(c)) == want and (lab[c[:, 1], c[:, 0]] == k).all()]
        assert match, k
        seen_outer.add(k)
    first = [int(np.lexsort((c[:, 0], c[:, 1]))[0]) for c in contours]
    starts = [(int(c[i, 1]), int(c[i, 0])) for c, i in zip(contours, first)]
    # (the raster-first pixel of a border is where an outer border was discovered; hole borders start next to the hole, so only
    #  the outer ones are checked for order)
    outer_starts = [s for c, s in zip(contours, starts) if len(set(O._hull(c))) and any(
        set(O._hull(c)) == set(O._hull(np.stack(np.nonzero(lab == lab[c[0, 1], c[0, 0]])[::-1], 1))) for _ in [0])]
    assert outer_starts == sorted(outer_starts, reverse=True)


def test_contour_of_simple_shapes():
    bm = np.zeros((7, 9), bool)
    bm[2:5, 3:7] = True                                   # 3 x 4 rectangle
    (c,) = O.find_contours(bm)
    assert set(map(tuple, c.tolist())) == {(x, y) for x in range(3, 7) for y in range(2, 5)} - {(4, 3), (5, 3)}
    bm[3, 4] = False                                      # a one-pixel hole: a second border around it, listed first
    cs = O.find_contours(bm)
    assert len(cs) == 2
    assert set(map(tuple, cs[0].tolist())) == {(4, 2), (3, 3), (5, 3), (4, 4)}          # the 4-neighbours of the hole
    single = np.zeros((3, 3), bool)
    single[1, 1] = True
    (c,) = O.find_contours(single)
    assert c.tolist() == [[1, 1]]


def _inside_or_on(poly, x, y, eps=1e-9):
    s = []
    for a in range(len(poly)):
        (x0, y0), (x1, y1) = poly[a - 1], poly[a]
        s.append((x1 - x0) * (y - y0) - (y1 - y0) * (x - x0))
    return all(v >= -eps for v in s) or all(v <= eps for v in s)


@pytest.mark.parametrize("seed", range(6))
def test_fill_poly_rule(seed):
    """axis-aligned boxes fill exactly their inclusive pixel rectangle; for any convex quadrilateral the mask contains every pixel
    whose centre lies inside or on the polygon and nothing farther than one pixel (Chebyshev) from such a pixel; the vertices
    are always set"""
    rng = np.random.default_rng(seed)
    m = np.zeros((20, 30), np.uint8)
    O.fill_poly(m, np.array([[4, 3], [17, 3], [17, 9], [4, 9]]))
    want = np.zeros_like(m)
    want[3:10, 4:18] = 1
    assert np.array_equal(m, want)
    for _ in range(40):
        cx, cy, hw, hh, th = rng.uniform(12, 50), rng.uniform(12, 40), rng.uniform(2, 11), rng.uniform(1, 8), rng.uniform(-1.6, 1.6)
        u, v = np.array([math.cos(th), math.sin(th)]), np.array([-math.sin(th), math.cos(th)])
        quad = np.array([[cx, cy] - hw * u - hh * v, [cx, cy] + hw * u - hh * v, [cx, cy] + hw * u + hh * v, [cx, cy] - hw * u + hh * v]).astype(np.int32)
        m = np.zeros((56, 64), np.uint8)
        O.fill_poly(m, quad)
        inside = np.zeros_like(m, bool)
        for y in range(56):
            for x in range(64):
                inside[y, x] = _inside_or_on(quad.tolist(), x, y)
        assert (m[inside] == 1).all()
        near = scipy.ndimage.binary_dilation(inside, structure=np.ones((3, 3), bool))
        assert not (m.astype(bool) & ~near).any()
        assert all(m[y, x] == 1 for x, y in quad.tolist())


def test_line_pixels_have_a_closed_form():
    """the 8-connected line of the fill rule, pixel by pixel, equals y0 + sign * max(0, (2 * minor * k + major - 1) // (2 * major)) along its
    major axis (the device kernel tests membership with this formula instead of walking the line)"""
    rng = np.random.default_rng(0)
    for _ in range(300):
        x0, y0, x1, y1 = (int(v) for v in rng.integers(0, 40, 4))
        m = np.zeros((40, 40), np.uint8)
        O._line8(m, (x0, y0), (x1, y1))
        want = np.zeros_like(m)
        ax, ay, bx, by = (x0, y0, x1, y1) if x1 >= x0 else (x1, y1, x0, y0)
        dx, dy = bx - ax, abs(by - ay)
        sy = 1 if by >= ay else -1
        if dy > dx:
            for k in range(dy + 1):
                want[ay + sy * k, ax + max(0, (2 * dx * k + dy - 1) // (2 * dy))] = 1
        else:
            for k in range(dx + 1):
                want[ay + sy * (max(0, (2 * dy * k + dx - 1) // (2 * dx)) if dx else 0), ax + k] = 1
        assert np.array_equal(m, want), (x0, y0, x1, y1)


@pytest.mark.parametrize("seed", range(4))
def test_clipper_offset_definition(seed):
    """every vertex of the offset polygon lies at the offset distance from the source polygon (within the integer rounding and the
    0.25 arc tolerance), the polygon encloses the source, and an axis-aligned rectangle grows by round(distance) on every side"""
    rng = np.random.default_rng(seed)
    out = O.clipper_offset_round([(10, 20), (110, 20), (110, 50), (10, 50)], 11.4)
    xs, ys = [p[0] for p in out], [p[1] for p in out]
    assert (min(xs), max(xs), min(ys), max(ys)) == (-1, 121, 9, 61)
    out_r = O.clipper_offset_round([(10, 50), (110, 50), (110, 20), (10, 20)], 11.4)      # the other orientation: same polygon
    assert sorted(out) == sorted(out_r)
    for _ in range(30):
        cx, cy, hw, hh, th, d = rng.uniform(100, 300), rng.uniform(100, 300), rng.uniform(5, 90), rng.uniform(3, 20), rng.uniform(-1.5, 1.5), rng.uniform(2, 30)
        u, v = np.array([math.cos(th), math.sin(th)]), np.array([-math.sin(th), math.cos(th)])
        quad = np.array([[cx, cy] - hw * u - hh * v, [cx, cy] + hw * u - hh * v, [cx, cy] + hw * u + hh * v, [cx, cy] - hw * u + hh * v])
        src = np.trunc(quad)
        out = np.array(O.clipper_offset_round(quad.tolist(), d), np.float64)
        assert len(out) >= 8

        def dist_to_poly(p):
            best = 1e9
            for a in range(4):
                a0, a1 = src[a - 1], src[a]
                t = np.clip(np.dot(p - a0, a1 - a0) / max(np.dot(a1 - a0, a1 - a0), 1e-12), 0, 1)
                best = min(best, np.linalg.norm(p - (a0 + t * (a1 - a0))))
            return best

        dd = np.array([dist_to_poly(p) for p in out])
        assert dd.max() <= d + 0.75 and dd.min() >= d - 0.25 - 0.75, (dd.min(), dd.max(), d)


@pytest.mark.parametrize("seed", range(5))
def test_min_area_rect_definition(seed):
    rng = np.random.default_rng(seed)
    pts = rng.integers(0, 80, size=(int(rng.integers(3, 40)), 2))
    corners, (w, h) = O.min_area_rect(pts)
    c = corners.astype(np.float64)
    e0, e1 = c[1] - c[0], c[3] - c[0]
    assert abs(np.dot(e0, e1)) <= 1e-3 * max(np.linalg.norm(e0) * np.linalg.norm(e1), 1.0)
    assert abs(np.linalg.norm(e0) - w) <= 1e-3 and abs(np.linalg.norm(e1) - h) <= 1e-3
    u, v = e0 / max(np.linalg.norm(e0), 1e-12), e1 / max(np.linalg.norm(e1), 1e-12)
    pu, pv = (pts - c[0]) @ u, (pts - c[0]) @ v
    assert pu.min() >= -1e-3 and pu.max() <= w + 1e-3 and pv.min() >= -1e-3 and pv.max() <= h + 1e-3
    for th in np.linspace(0, math.pi / 2, 721):           # no orientation of a 0.125-degree grid does better
        ru, rv = pts @ np.array([math.cos(th), math.sin(th)]), pts @ np.array([-math.sin(th), math.cos(th)])
        assert (np.ptp(ru)) * (np.ptp(rv)) >= w * h - 1e-6 * max(w * h, 1)


def test_box_score_and_pipeline_on_an_axis_aligned_blob():
    """a solid axis-aligned blob: the score is the blob's probability, the box is the blob grown by round(unclip distance) and
    scaled to the source image"""
    prob = np.full((60, 120), 0.1, np.float32)
    prob[20:31, 30:91] = 0.8                              # 11 rows x 61 columns: corners (30,20)..(90,30)
    boxes, scores = O.db_postprocess(prob, 120, 240)
    assert len(scores) == 1 and abs(scores[0] - 0.8) < 1e-6
    d = (60 * 10) * 1.5 / (2 * (60 + 10))                 # area * ratio / perimeter of the 60 x 10 rectangle of pixel centres
    g = int(d + 0.5)
    want = np.array([[30 - g, 20 - g], [90 + g, 20 - g], [90 + g, 30 + g], [30 - g, 30 + g]]) * 2
    assert np.array_equal(boxes[0], want)
    prob[20:31, 30:91] = 0.59                             # below box_thresh
    assert len(O.db_postprocess(prob, 120, 240)[1]) == 0
    prob[:] = 0.1
    prob[20:22, 30:91] = 0.9                              # short side 1 < min_size 3
    assert len(O.db_postprocess(prob, 120, 240)[1]) == 0


# ---------------------------------------------------------------------------------------------------------------- part 2
@pytest.mark.parametrize("seed,H,W,nboxes,holes,specks", [(1, 544, 960, 6, 0, 0), (2, 544, 960, 40, 0, 10), (3, 96, 160, 3, 3, 4),
                                                          (4, 544, 960, 0, 0, 0), (5, 272, 480, 12, 8, 20), (6, 544, 960, 25, 5, 5)])
def test_product_host_postprocess_equals_the_oracle(seed, H, W, nboxes, holes, specks):
    import vsr_amd  # noqa: F401
    from vsr_amd.backend.tools import ocr_det

    prob = blob_map(seed, H, W, nboxes, holes, specks)
    want_b, want_s = O.db_postprocess(prob, 1080, 1920)
    got_b, got_s = ocr_det.db_postprocess(prob, 1080, 1920)
    assert got_b.shape == want_b.shape, (got_b.shape, want_b.shape)
    assert np.array_equal(got_b.astype(np.int64), want_b.astype(np.int64))
    assert np.allclose(got_s, want_s, rtol=0, atol=1e-6)
    if nboxes >= 6:
        assert len(want_s) >= 3


def test_product_host_postprocess_max_candidates_and_ratio():
    import vsr_amd  # noqa: F401
    from vsr_amd.backend.tools import ocr_det

    prob = blob_map(8, 272, 480, 14, 2, 30)
    for kw in (dict(max_candidates=5), dict(unclip_ratio=2.0, box_thresh=0.5), dict(thresh=0.5)):
        want_b, want_s = O.db_postprocess(prob, 544, 960, **kw)
        got_b, got_s = ocr_det.db_postprocess(prob, 544, 960, **kw)
        assert np.array_equal(got_b.astype(np.int64), want_b.astype(np.int64)) and np.allclose(got_s, want_s, atol=1e-6)
