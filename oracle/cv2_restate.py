"""numpy restatement of the OpenCV calls on the STTN path (oracle; PARITY UNPINNED, see __init__.py).

opencv-python==4.11.0.86 (reference requirements.txt:2) is not available here, so these follow
the published OpenCV 4.11 algorithm (modules/imgproc/src/resize.cpp, resize() generic path,
INTER_LINEAR) and are anchored on the reference's call sites:
  cv2.resize(image_crop, (640, 120))            sttn_auto_inpaint.py:271  (uint8, fixed point)
  cv2.resize(comps[k][j], (W_ori, split_h))     sttn_auto_inpaint.py:312  (uint8 or float32)
  cv2.threshold(mask, 127, 1, THRESH_BINARY)    sttn_auto_inpaint.py:224
  cv2.rectangle(mask, (x1,y1), (x2,y2), 255,-1) tools/inpaint_tools.py:45
  cv2.connectedComponentsWithStats(.., 8)       tools/inpaint_tools.py:77
"""
import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def linear_tables(ssize, dsize, clamp_x):
    """resize(): per-destination source offset and the two taps (float and x2048 short).

    fx = (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx; horizontally sx is clamped
    to [0, ssize-1] with fx reset to 0; vertically fy is kept and the rows are clipped at use.
    saturate_cast<short>(float) rounds half to even (cvRound).
    """
    inv_scale = float(dsize) / float(ssize)
    scale = 1.0 / inv_scale
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_x:
        lo = s < 0
        f[lo] = 0
        s[lo] = 0
        hi = s >= ssize - 1
        f[hi] = 0
        s[hi] = ssize - 1
    c0 = (np.float32(1.0) - f).astype(np.float32)
    c1 = f
    fcoef = np.stack([c0, c1], axis=1).astype(np.float32)
    icoef = np.clip(np.rint(fcoef * np.float32(INTER_RESIZE_COEF_SCALE)), -32768, 32767).astype(np.int16)
    return s, icoef, fcoef


def resize_linear(img, dsize):
    """cv2.resize(img, (dw, dh)) with the default INTER_LINEAR for HxWxC uint8 or float32."""
    dw, dh = dsize
    sh, sw = img.shape[:2]
    xofs, ialpha, falpha = linear_tables(sw, dw, True)
    yofs, ibeta, fbeta = linear_tables(sh, dh, False)
    x0 = xofs
    x1 = np.minimum(x0 + 1, sw - 1)
    y0 = np.clip(yofs, 0, sh - 1)
    y1 = np.clip(yofs + 1, 0, sh - 1)
    if img.dtype == np.uint8:
        # HResizeLinear<uchar,int,short> then VResizeLinear<uchar,int,short,FixedPtCast<..,22>>:
        # dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
        # (exact 2x down-scaling is routed to INTER_AREA by resize(); the result is identical)
        s = img.astype(np.int32)
        a0 = ialpha[:, 0].astype(np.int32)[None, :, None]
        a1 = ialpha[:, 1].astype(np.int32)[None, :, None]
        b0 = ibeta[:, 0].astype(np.int32)[:, None, None]
        b1 = ibeta[:, 1].astype(np.int32)[:, None, None]
        r0, r1 = s[y0], s[y1]
        h0 = r0[:, x0] * a0 + r0[:, x1] * a1
        h1 = r1[:, x0] * a0 + r1[:, x1] * a1
        v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2
        return np.ascontiguousarray(np.clip(v, 0, 255).astype(np.uint8))
    if img.dtype == np.float32:
        s = img
        a0 = falpha[:, 0][None, :, None]
        a1 = falpha[:, 1][None, :, None]
        b0 = fbeta[:, 0][:, None, None]
        b1 = fbeta[:, 1][:, None, None]
        r0, r1 = s[y0], s[y1]
        h0 = (r0[:, x0] * a0).astype(np.float32) + (r0[:, x1] * a1).astype(np.float32)
        h1 = (r1[:, x0] * a0).astype(np.float32) + (r1[:, x1] * a1).astype(np.float32)
        return np.ascontiguousarray(((h0 * b0).astype(np.float32) + (h1 * b1).astype(np.float32)).astype(np.float32))
    raise TypeError(f"resize_linear: unsupported dtype {img.dtype}")


def threshold_binary(mask, thresh, maxval):
    """cv2.threshold(mask, thresh, maxval, THRESH_BINARY)[1] for uint8."""
    return np.where(mask > thresh, np.uint8(maxval), np.uint8(0)).astype(np.uint8)


def rectangle_filled(mask, pt1, pt2, value):
    """cv2.rectangle(mask, pt1, pt2, value, thickness=-1): inclusive corners, clipped to the image."""
    (x1, y1), (x2, y2) = pt1, pt2
    xa, xb = min(x1, x2), max(x1, x2)
    ya, yb = min(y1, y2), max(y1, y2)
    h, w = mask.shape[:2]
    xa, ya = max(xa, 0), max(ya, 0)
    xb, yb = min(xb, w - 1), min(yb, h - 1)
    if xa <= xb and ya <= yb:
        mask[ya:yb + 1, xa:xb + 1] = value
    return mask


def connected_components_with_stats(binary, connectivity=8):
    """cv2.connectedComponentsWithStats: (num_labels, labels, stats[left,top,width,height,area], centroids)."""
    from scipy import ndimage

    structure = np.ones((3, 3), dtype=bool) if connectivity == 8 else None
    labels, n = ndimage.label(binary > 0, structure=structure)
    stats = np.zeros((n + 1, 5), dtype=np.int32)
    cents = np.zeros((n + 1, 2), dtype=np.float64)
    for i in range(n + 1):
        ys, xs = np.nonzero(labels == i)
        if ys.size == 0:
            continue
        stats[i] = (xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, ys.size)
        cents[i] = (xs.mean(), ys.mean())
    return n + 1, labels.astype(np.int32), stats, cents
