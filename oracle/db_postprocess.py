"""CPU restatement of the text detector's post-process, DBPostProcess (TEST INFRASTRUCTURE ONLY: imported by tests/ alone).

The reference hands every sampled frame to paddleocr's `TextDetection.predict` (backend/tools/subtitle_detect.py:41-58) and
reads `res['dt_polys']` (:61-63).  What turns the network's probability map into those quadrilaterals is PaddleX's
`DBPostProcess` (paddlex/inference/models/text_detection/processors.py in paddlex 3.x, the package paddleocr==3.4.0 --
requirements.txt:9 -- delegates to), configured by backend/models/V5/ch_det/inference.yml:49-53
(thresh 0.3, box_thresh 0.6, max_candidates 1000, unclip_ratio 1.5; box_type "quad", score_mode "fast", no dilation are the
class defaults the yml leaves alone).  Neither paddlex nor its two native dependencies for this step -- opencv-python 4.11
(findContours, minAreaRect, boxPoints, fillPoly, mean, contourArea, arcLength) and pyclipper 1.3 (ClipperLib 6.4.2 polygon
offsetting) -- are in the image or the mount, so PARITY IS UNPINNED: this file restates the published algorithms

    process():             bitmap = pred > thresh; boxes_from_bitmap(pred, bitmap, src_w, src_h)
    boxes_from_bitmap():   contours = cv2.findContours(bitmap * 255, RETR_LIST, CHAIN_APPROX_SIMPLE)  (first max_candidates)
                           per contour: get_mini_boxes -> skip if short side < 3; box_score_fast -> skip if < box_thresh;
                           unclip; get_mini_boxes -> skip if short side < 5; scale to the source image, round, clip; int16
    get_mini_boxes():      cv2.minAreaRect + cv2.boxPoints, corners ordered (top-left, top-right, bottom-right, bottom-left)
    box_score_fast():      mean of pred over cv2.fillPoly(mask, box.astype(int32)) inside the box's bounding rows / columns
    unclip():              pyclipper offset (JT_ROUND, ET_CLOSEDPOLYGON) by area * unclip_ratio / perimeter

and each native primitive from ITS published algorithm (cited at the function).  Independent of the product's
backend/tools/ocr_det.py: nothing here is imported from or shared with it.  tests/test_db_postprocess.py holds the primitives to
definitions they must satisfy whatever the implementation (contour <-> connected component correspondence, fill rule on
axis-aligned boxes, offset distance, minimal area) and the product -- host and device -- to this file.
"""
import math

import numpy as np

F32 = np.float32


# ------------------------------------------------------------------------------------------------------------------------
# cv2.findContours(img, RETR_LIST, CHAIN_APPROX_SIMPLE): Suzuki & Abe 1985, "Topological structural analysis of digitized binary
# images by border following" (the algorithm OpenCV documents for this function), 8-connected foreground.  RETR_LIST keeps
# every border -- outer borders AND hole borders -- without hierarchy.  Points are (x, y).  CHAIN_APPROX_SIMPLE drops the
# interior points of straight runs; this restatement keeps every border pixel, which changes nothing downstream: the only
# consumer is minAreaRect, a function of the convex hull.
# Order: OpenCV returns the borders in REVERSE order of discovery by the raster scan (the last border found comes first;
# observed behaviour of cv2.findContours, kept by the 4.x reimplementation -- unverifiable here).
# ------------------------------------------------------------------------------------------------------------------------
_NB8 = [(0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1)]    # (di, dj) counter-clockwise from east (image rows grow down)


def find_contours(bitmap):
    """bitmap [H,W] bool / 0-1 -> list of int32 arrays [n,2] of (x, y) border points, in cv2's order"""
    H, W = bitmap.shape
    f = np.zeros((H + 2, W + 2), np.int32)                # the frame is background, as in cv2 (it pads the image itself)
    f[1:-1, 1:-1] = (np.asarray(bitmap) != 0)
    nbd = 1
    found = []
    # candidate start pixels of the raster scan (the marks written while following can only remove candidates)
    cand = np.argwhere((f != 0) & ((np.roll(f, 1, axis=1) == 0) | (np.roll(f, -1, axis=1) == 0)))
    for i, j in cand:
        i, j = int(i), int(j)
        if f[i, j] == 1 and f[i, j - 1] == 0:             # (1a) outer border starts
            start_nb = 4                                  # the pixel to the west
        elif f[i, j] >= 1 and f[i, j + 1] == 0:           # (1b) hole border starts
            start_nb = 0                                  # the pixel to the east
        else:
            continue
        nbd += 1
        pts = []
        # (3.1) clockwise from (i2, j2) around (i, j): first non-zero pixel
        first = None
        for k in range(8):
            d = (start_nb - k) % 8
            if f[i + _NB8[d][0], j + _NB8[d][1]] != 0:
                first = d
                break
        if first is None:                                 # a single pixel
            f[i, j] = -nbd
            found.append(np.array([[j - 1, i - 1]], np.int32))
            continue
        i1, j1 = i + _NB8[first][0], j + _NB8[first][1]
        i2, j2, i3, j3 = i1, j1, i, j
        while True:
            # (3.3) counter-clockwise around (i3, j3), starting after (i2, j2): first non-zero pixel (i4, j4)
            d0 = _NB8.index((i2 - i3, j2 - j3))
            east_zero_seen = False
            for k in range(1, 9):
                d = (d0 + k) % 8
                ii, jj = i3 + _NB8[d][0], j3 + _NB8[d][1]
                if f[ii, jj] != 0:
                    i4, j4 = ii, jj
                    break
                if d == 0:
                    east_zero_seen = True                 # (i3, j3 + 1) is a 0-pixel examined in this step
            # (3.4)
            if east_zero_seen:
                f[i3, j3] = -nbd
            elif f[i3, j3] == 1:
                f[i3, j3] = nbd
            pts.append((j3 - 1, i3 - 1))
            # (3.5)
            if (i4, j4) == (i, j) and (i3, j3) == (i1, j1):
                break
            i2, j2, i3, j3 = i3, j3, i4, j4
        found.append(np.array(pts, np.int32))
    return found[::-1]


# ------------------------------------------------------------------------------------------------------------------------
# cv2.minAreaRect / cv2.boxPoints (imgproc/src/rotcalipers.cpp, cv::minAreaRect; imgproc/src/drawing? no: RotatedRect::points):
# the rectangle of minimum area over the directions of the convex hull's edges ("rotating calipers").
# OpenCV works in float32; this restatement computes in float64 and casts the four corners to float32 as boxPoints returns them.
# ------------------------------------------------------------------------------------------------------------------------
def _hull(pts):
    p = sorted(set((float(x), float(y)) for x, y in pts))
    if len(p) <= 2:
        return p

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lo, up = [], []
    for q in p:
        while len(lo) >= 2 and cross(lo[-2], lo[-1], q) <= 0:
            lo.pop()
        lo.append(q)
    for q in reversed(p):
        while len(up) >= 2 and cross(up[-2], up[-1], q) <= 0:
            up.pop()
        up.append(q)
    return lo[:-1] + up[:-1]


def min_area_rect(points):
    """-> (corners float32 [4,2] in boxPoints' cyclic order, (w, h) of the rectangle)"""
    h = _hull(np.asarray(points).reshape(-1, 2))
    if len(h) == 1:
        return np.array([h[0]] * 4, F32), (0.0, 0.0)
    if len(h) == 2:                                       # cv::minAreaRect, n == 2: width = the distance, height 0
        (x0, y0), (x1, y1) = h
        return np.array([h[0], h[0], h[1], h[1]], F32), (math.hypot(x1 - x0, y1 - y0), 0.0)
    best = None
    n = len(h)
    for a in range(n):
        (x0, y0), (x1, y1) = h[a], h[(a + 1) % n]
        ln = math.hypot(x1 - x0, y1 - y0)
        ux, uy = (x1 - x0) / ln, (y1 - y0) / ln
        pu = [x * ux + y * uy for x, y in h]
        pv = [-x * uy + y * ux for x, y in h]
        w, hh = max(pu) - min(pu), max(pv) - min(pv)
        if best is None or w * hh < best[0]:              # strict '<': the first minimum of the scan
            best = (w * hh, ux, uy, min(pu), max(pu), min(pv), max(pv))
    _, ux, uy, u0, u1, v0, v1 = best
    vx, vy = -uy, ux
    c = [(u0 * ux + v0 * vx, u0 * uy + v0 * vy), (u1 * ux + v0 * vx, u1 * uy + v0 * vy),
         (u1 * ux + v1 * vx, u1 * uy + v1 * vy), (u0 * ux + v1 * vx, u0 * uy + v1 * vy)]
    return np.array(c, F32), (u1 - u0, v1 - v0)


def get_mini_boxes(contour):
    """DBPostProcess.get_mini_boxes: (corners [4,2] float32 ordered top-left, top-right, bottom-right, bottom-left; short side)"""
    corners, (w, h) = min_area_rect(contour)
    p = sorted(corners.tolist(), key=lambda q: q[0])      # Python's sort is stable, like sorted() in the original
    i1, i4 = (0, 1) if p[1][1] > p[0][1] else (1, 0)
    i2, i3 = (2, 3) if p[3][1] > p[2][1] else (3, 2)
    return np.array([p[i1], p[i2], p[i3], p[i4]], F32), min(w, h)


# ------------------------------------------------------------------------------------------------------------------------
# cv2.fillPoly (imgproc/src/drawing.cpp: CollectPolyEdges + FillEdgeCollection, line_type LINE_8, shift 0) with integer
# vertices: every polygon edge is DRAWN as an 8-connected line (LineIterator, left to right), and every scan line y in
# [y_top, y_bottom) of an edge pair is filled from floor(x_left) to floor(x_right) in 16.16 fixed point, the x of an edge advancing
# by the truncated quotient dx = (x1 - x0) / (y1 - y0) per row.
# ------------------------------------------------------------------------------------------------------------------------
def _line8(mask, p0, p1):
    (x0, y0), (x1, y1) = p0, p1
    dx, dy = x1 - x0, y1 - y0
    if dx < 0:                                            # leftToRight: start from the left end point
        x0, y0, dx, dy = x1, y1, -dx, -dy
    sy = -1 if dy < 0 else 1
    dy = abs(dy)
    steep = dy > dx
    major, minor = (dy, dx) if steep else (dx, dy)
    err = major - 2 * minor
    x, y = x0, y0
    H, W = mask.shape
    for _ in range(major + 1):
        if 0 <= y < H and 0 <= x < W:
            mask[y, x] = 1
        step_minor = err < 0
        err += (2 * major - 2 * minor) if step_minor else (-2 * minor)
        if steep:
            y += sy
            x += 1 if step_minor else 0
        else:
            x += 1
            y += sy if step_minor else 0


def fill_poly(mask, pts):
    """mask [H,W] uint8 (modified), pts int [n,2] (x, y): cv2.fillPoly(mask, [pts], 1)"""
    H, W = mask.shape
    n = len(pts)
    edges = []
    for a in range(n):
        x0, y0 = int(pts[a - 1][0]), int(pts[a - 1][1])
        x1, y1 = int(pts[a][0]), int(pts[a][1])
        _line8(mask, (x0, y0), (x1, y1))
        if y0 == y1:
            continue
        fx0, fx1 = x0 << 16, x1 << 16
        q = abs(fx1 - fx0) // abs(y1 - y0)               # C++ integer division truncates toward zero
        dx = q if (fx1 - fx0 >= 0) == (y1 - y0 > 0) else -q
        edges.append([y0, y1, fx0, dx] if y0 < y1 else [y1, y0, fx1, dx])
    if len(edges) < 2:
        return mask
    y_min, y_max = min(e[0] for e in edges), min(max(e[1] for e in edges), H)
    for y in range(y_min, y_max):
        xs = sorted(e[2] + (y - e[0]) * e[3] for e in edges if e[0] <= y < e[1])
        if y < 0:
            continue
        for a in range(0, len(xs) - 1, 2):
            xa, xb = xs[a] >> 16, xs[a + 1] >> 16
            if xa < W and xb >= 0:
                mask[y, max(xa, 0):min(xb, W - 1) + 1] = 1
    return mask


def box_score_fast(pred, box):
    """DBPostProcess.box_score_fast: pred [H,W] float32, box [4,2] float32"""
    h, w = pred.shape
    box = np.array(box, F32)
    xmin = max(0, min(math.floor(box[:, 0].min()), w - 1))
    xmax = max(0, min(math.ceil(box[:, 0].max()), w - 1))
    ymin = max(0, min(math.floor(box[:, 1].min()), h - 1))
    ymax = max(0, min(math.ceil(box[:, 1].max()), h - 1))
    mask = np.zeros((ymax - ymin + 1, xmax - xmin + 1), np.uint8)
    box[:, 0] -= F32(xmin)
    box[:, 1] -= F32(ymin)
    fill_poly(mask, box.astype(np.int32))                # astype truncates toward zero
    sel = mask.astype(bool)
    if not sel.any():
        return 0.0                                        # cv2.mean of an empty mask
    return float(pred[ymin:ymax + 1, xmin:xmax + 1][sel].astype(np.float64).mean())


# ------------------------------------------------------------------------------------------------------------------------
# pyclipper.PyclipperOffset().AddPath(box, JT_ROUND, ET_CLOSEDPOLYGON); Execute(distance): ClipperLib 6.4.2 (clipper.cpp,
# ClipperOffset::AddPath / FixOrientations / DoOffset / OffsetPoint / DoRound; ArcTolerance 0.25, MiterLimit 2).  pyclipper takes
# integer coordinates: the float corners are truncated on the way in (Cython's conversion to long long).  Every output vertex is
# rounded to an integer (Round(): half away from zero).  The final union that Execute() runs over the offset polygon only removes
# collinear vertices of this convex case, which no consumer here can see (minAreaRect again).
# ------------------------------------------------------------------------------------------------------------------------
def _cround(v):
    return int(v - 0.5) if v < 0 else int(v + 0.5)


def clipper_offset_round(path, delta):
    """path: sequence of (x, y) numbers, closed polygon -> list of integer (x, y) of the polygon offset by delta > 0"""
    src = []
    for x, y in path:
        q = (int(x), int(y))
        if not src or q != src[-1]:
            src.append(q)
    if len(src) > 1 and src[0] == src[-1]:
        src.pop()
    n = len(src)
    if n < 3:
        return []
    area2 = sum((src[j - 1][0] + src[j][0]) * (src[j - 1][1] - src[j][1]) for j in range(n))
    if -area2 * 0.5 < 0:                                  # Orientation() false: the path is reversed so that the normals point outwards
        src.reverse()
    y = min(0.25, abs(delta) * 0.25)
    steps = math.pi / math.acos(1 - y / abs(delta))
    steps = min(steps, abs(delta) * math.pi)
    m_sin, m_cos, steps_per_rad = math.sin(2 * math.pi / steps), math.cos(2 * math.pi / steps), steps / (2 * math.pi)
    normals = []
    for j in range(n):
        (x0, y0), (x1, y1) = src[j], src[(j + 1) % n]
        dx, dy = x1 - x0, y1 - y0
        f = 1.0 / math.sqrt(dx * dx + dy * dy)
        normals.append((dy * f, -dx * f))
    out = []
    k = n - 1
    for j in range(n):
        sin_a = normals[k][0] * normals[j][1] - normals[j][0] * normals[k][1]
        done = False
        if abs(sin_a * delta) < 1.0:
            cos_a = normals[k][0] * normals[j][0] + normals[j][1] * normals[k][1]
            if cos_a > 0:
                out.append((_cround(src[j][0] + normals[k][0] * delta), _cround(src[j][1] + normals[k][1] * delta)))
                done = True
        else:
            sin_a = max(-1.0, min(1.0, sin_a))
        if not done:
            if sin_a * delta < 0:                         # concave corner
                out.append((_cround(src[j][0] + normals[k][0] * delta), _cround(src[j][1] + normals[k][1] * delta)))
                out.append(src[j])
                out.append((_cround(src[j][0] + normals[j][0] * delta), _cround(src[j][1] + normals[j][1] * delta)))
            else:                                         # DoRound
                a = math.atan2(sin_a, normals[k][0] * normals[j][0] + normals[k][1] * normals[j][1])
                nst = max(_cround(steps_per_rad * abs(a)), 1)
                X, Y = normals[k]
                for _ in range(nst):
                    out.append((_cround(src[j][0] + X * delta), _cround(src[j][1] + Y * delta)))
                    X, Y = X * m_cos - m_sin * Y, X * m_sin + Y * m_cos
                out.append((_cround(src[j][0] + normals[j][0] * delta), _cround(src[j][1] + normals[j][1] * delta)))
        k = j
    return out


def unclip(box, unclip_ratio):
    """DBPostProcess.unclip: box [4,2] float32 -> integer points of the expanded polygon"""
    b = np.asarray(box, F32)
    x, y = b[:, 0].astype(np.float64), b[:, 1].astype(np.float64)
    area = abs(float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))) * 0.5            # cv2.contourArea
    d = b - np.roll(b, 1, axis=0)                                                       # cv2.arcLength(closed): float32 segment lengths
    length = float(np.sum(np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(F32)).astype(np.float64)))
    if length == 0.0 or area == 0.0:
        return np.zeros((0, 2), np.int64)
    return np.array(clipper_offset_round(b.tolist(), area * unclip_ratio / length), np.int64).reshape(-1, 2)


# ------------------------------------------------------------------------------------------------------------------------
def db_postprocess(pred, src_h, src_w, thresh=0.3, box_thresh=0.6, max_candidates=1000, unclip_ratio=1.5, min_size=3):
    """DBPostProcess.process + boxes_from_bitmap (box_type 'quad', score_mode 'fast', no dilation).
    pred [H,W] float32 probability map, (src_h, src_w) the source image -> (boxes int16 [n,4,2], scores list)"""
    pred = np.asarray(pred, F32)
    height, width = pred.shape
    width_scale, height_scale = src_w / width, src_h / height
    contours = find_contours(pred > thresh)
    boxes, scores = [], []
    for contour in contours[:max_candidates]:
        points, sside = get_mini_boxes(contour)
        if sside < min_size:
            continue
        score = box_score_fast(pred, points.reshape(-1, 2))
        if box_thresh > score:
            continue
        expanded = unclip(points, unclip_ratio)
        if len(expanded) == 0:
            continue
        box, sside = get_mini_boxes(expanded.reshape(-1, 2))
        if sside < min_size + 2:
            continue
        out = np.zeros((4, 2), np.float64)
        for i in range(4):                                # Python round(): half to even
            out[i, 0] = max(0, min(round(float(box[i][0]) * width_scale), src_w))
            out[i, 1] = max(0, min(round(float(box[i][1]) * height_scale), src_h))
        boxes.append(out.astype(np.int16))
        scores.append(score)
    return (np.array(boxes, np.int16).reshape(-1, 4, 2), scores)
