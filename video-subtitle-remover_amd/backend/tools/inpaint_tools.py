"""Host bookkeeping of the reference's backend/tools/inpaint_tools.py, cv2-free.

These decide WHICH pixels are inpainted and how frames are batched (so they change pixels):
create_mask (:31-47), get_inpaint_area_by_mask (:49-242), batch_generator (:7-29),
is_frame_number_in_ab_sections (:303-324).  OpenCV calls are replaced by numpy / scipy.ndimage
equivalents (filled rectangle with inclusive corners, 8-connected components with stats).
"""
import numpy as np

from ..config import config


def batch_generator(data, max_batch_size):
    n_samples = len(data)
    batch_size = max_batch_size
    num_batches = n_samples // batch_size
    # the reference keeps shrinking while the remainder is under half a batch -- remainder 0 included
    while n_samples % batch_size < batch_size / 2.0 and batch_size > 1:
        batch_size -= 1
        num_batches = n_samples // batch_size
    for i in range(num_batches):
        yield data[i * batch_size:(i + 1) * batch_size]
    tail = num_batches * batch_size
    if tail < n_samples:
        yield data[tail:]


def _fill_rect(mask, x1, y1, x2, y2, value):
    h, w = mask.shape[:2]
    xa, xb = max(min(x1, x2), 0), min(max(x1, x2), w - 1)
    ya, yb = max(min(y1, y2), 0), min(max(y1, y2), h - 1)
    if xa <= xb and ya <= yb:
        mask[ya:yb + 1, xa:xb + 1] = value


def create_mask(size, coords_list):
    mask = np.zeros(size, dtype="uint8")
    grow = config.subtitleAreaDeviationPixel.value
    if coords_list:
        for xmin, xmax, ymin, ymax in coords_list:
            _fill_rect(mask, max(xmin - grow, 0), max(ymin - grow, 0), xmax + grow, ymax + grow, 255)
    return mask


def threshold_mask(input_mask):
    """cv2.threshold(input_mask, 127, 1, cv2.THRESH_BINARY)[1][:, :, None] (sttn_auto_inpaint.py:224-225)."""
    return (input_mask > 127).astype(np.uint8)[:, :, None]


def _islands(binary):
    """8-connected components of the mask as (top, bottom, centre row, area, label) -- the statistics the reference reads off
    cv2.connectedComponentsWithStats (inpaint_tools.py:151-160), labels numbered in raster order of their first pixel.

    Run-based: a subtitle mask is a union of a few rectangles, i.e. one to three runs of set pixels per row; the components are
    the classes of a union-find over those runs (two runs of neighbouring rows belong together when they touch or overlap by a
    corner), and top / bottom / area / row sum add up per run.  scipy.ndimage.label does the same per pixel -- importing it cost
    90 ms in front of the first chunk (profiles/r04_cli_startup.log); it stays the path for masks with a great many runs."""
    b = np.asarray(binary) > 0
    H, W = b.shape
    edge = np.diff(b.astype(np.int8), axis=1, prepend=0, append=0)          # [H, W + 1]: +1 where a run starts, -1 one past its end
    ry, rs = np.nonzero(edge == 1)
    _, re_ = np.nonzero(edge == -1)                                          # row-major on both sides: the k-th start meets the k-th end
    n = int(ry.size)
    if n == 0:
        return []
    if n > 50000:
        return _islands_scipy(b)
    ry, rs, re_ = ry.tolist(), rs.tolist(), re_.tolist()
    parent = list(range(n))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    first = {}                                                               # row -> index of its first run
    for k, y in enumerate(ry):
        first.setdefault(y, k)
    for k in range(n):
        y = ry[k]
        j = first.get(y - 1)
        if j is None:
            continue
        while j < n and ry[j] == y - 1:
            if rs[j] <= re_[k] and rs[k] <= re_[j]:                          # [rs - 1, re + 1) overlaps: 8-connectivity
                a, c = find(j), find(k)
                if a != c:
                    parent[max(a, c)] = min(a, c)                            # the root is the component's first run in raster order
            j += 1
    comps = {}
    for k in range(n):
        r = find(k)
        ln = re_[k] - rs[k]
        c = comps.get(r)
        if c is None:
            comps[r] = [ry[k], ry[k] + 1, ln, ln * ry[k]]
        else:
            if ry[k] + 1 > c[1]:
                c[1] = ry[k] + 1
            c[2] += ln
            c[3] += ln * ry[k]
    out = []
    for label, r in enumerate(sorted(comps), start=1):
        top, bottom, area, ysum = comps[r]
        if area < 10:
            continue
        cy = int(((ysum - area * top) / area) + top)                          # int(mean of the rows relative to the top + top), as before
        out.append((top, bottom, cy, area, label))
    return out


def _islands_scipy(b):
    from scipy import ndimage

    labels, n = ndimage.label(b, structure=np.ones((3, 3), dtype=bool))
    out = []
    if n == 0:
        return out
    objs = ndimage.find_objects(labels)
    for i, sl in enumerate(objs, start=1):
        if sl is None:
            continue
        ys, _ = np.nonzero(labels[sl] == i)
        area = int(ys.size)
        if area < 10:
            continue
        top, bottom = sl[0].start, sl[0].stop
        cy = int((ys.mean() + top))
        out.append((top, bottom, cy, area, i))
    return out


def get_inpaint_area_by_mask(W, H, h, mask, multiple=1):
    """-> [(ymin, ymax, xmin, xmax)] full-width strips of height exactly h covering the mask islands."""
    inpaint_area = []
    if np.all(mask == 0):
        return inpaint_area
    binary = ((mask > 0).astype(np.uint8) * 255)
    if binary.ndim == 3:
        binary = binary[:, :, 0]
    islands = _islands(binary)
    if not islands:
        return inpaint_area
    islands.sort(key=lambda t: t[2])
    merged, group = [], [islands[0]]
    for isl in islands[1:]:
        g_lo = min(t[0] for t in group)
        g_hi = max(t[1] for t in group)
        span = max(g_hi, isl[1]) - min(g_lo, isl[0])
        linked = True if g_hi >= isl[0] else bool(np.any(binary[g_hi:isl[0], :] > 0))
        if span <= h and linked:
            group.append(isl)
        else:
            merged.append(group)
            group = [isl]
    merged.append(group)

    def place(ymin):
        ymax = ymin + h
        if ymax > H:
            ymax, ymin = H, max(0, H - h)
        return ymin, ymax

    for group in merged:
        lo = min(t[0] for t in group)
        hi = max(t[1] for t in group)
        center = sum(t[2] for t in group) // len(group)
        ymin, ymax = place(max(0, center - h // 2))
        if ymin > lo or ymax < hi:
            ymin, ymax = place(lo) if hi - lo <= h else place(max(0, (lo + hi) // 2 - h // 2))
        xmin, xmax = 0, W
        if multiple > 1:
            height = ymax - ymin
            rem = height % multiple
            if rem:
                adj = multiple - rem
                mid = (ymin + ymax) / 2
                if ymin - adj / 2 >= 0 and ymax + adj / 2 <= H:
                    ymin, ymax = int(mid - height / 2 - adj / 2), int(mid + height / 2 + adj / 2)
                elif height > multiple:
                    ymin, ymax = int(mid - (height - rem) / 2), int(mid + (height - rem) / 2)
                elif ymax + adj <= H:
                    ymax += adj
                elif ymin - adj >= 0:
                    ymin -= adj
            rem_w = (xmax - xmin) % multiple
            if rem_w:
                width = xmax - xmin
                mid = (xmin + xmax) / 2
                xmin, xmax = int(mid - (width - rem_w) / 2), int(mid + (width - rem_w) / 2)
        area = (int(ymin), int(ymax), int(xmin), int(xmax))
        if area not in inpaint_area:
            inpaint_area.append(area)
    return inpaint_area


def expand_frame_ranges(frame_ranges, backward_frame_count, forward_frame_count):
    """tools/inpaint_tools.py:244-301 -- widen every (start, end) backwards/forwards without overlapping neighbours."""
    if not frame_ranges:
        return []
    ordered = sorted(frame_ranges)
    out = []
    for i, (start, end) in enumerate(ordered):
        new_start = max(1, start - backward_frame_count)
        new_end = end + forward_frame_count
        if i < len(ordered) - 1:
            next_start = ordered[i + 1][0]
            if new_end >= next_start:
                # contiguous neighbours keep their boundary, others stop one frame short of the neighbour
                new_end = end if next_start - end == 1 else min(new_end, next_start - 1)
        if out and new_start <= out[-1][1]:
            new_start = out[-1][1] + 1
        out.append((new_start, new_end) if new_start <= new_end else (start, end))
    return out


def is_frame_number_in_ab_sections(frame_no, ab_sections):
    if ab_sections is None or len(ab_sections) <= 0:
        return True
    return any(frame_no in section for section in ab_sections)
