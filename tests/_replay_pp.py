"""CPU replay of the ProPainter plans (test infrastructure): OP_EW sub-kinds of csrc/pp_plan.h executed with numpy / torch-CPU
exactly as csrc/pp_kernels.hip defines them."""
import ctypes as C

import numpy as np
import torch

import _replay

EW_PP_MASK_F32, EW_PP_IMGPROP, EW_PP_COPY = 30, 31, 32
PB_IN_FRAMES, PB_IN_MASK_U8, PB_IN_MASK_UPD_U8, PB_IN_FLOW_F, PB_IN_FLOW_B = 1, 2, 3, 4, 5
PB_FW, PB_FWM = 9, 10
BYTE_BUFS = (2, 3, 11)
OP_EW = 6


def imgprop_plan_view(_lib, t, H, W):
    p = C.c_void_p()
    _lib.check(_lib.lib.vsr_pp_imgprop_plan_create(t, H, W, C.byref(p)))
    return _replay.PlanView(_lib, None, 0, plan_ptr=p)


def _warp_coord(pos, size):
    d = np.float32(max(size - 1, 1))
    g = np.float32(2.0) * pos / d - np.float32(1.0)
    return ((g + np.float32(1.0)) / np.float32(2.0)) * np.float32(size - 1)


def _bilinear(img, iy, ix):
    h, w = img.shape
    y0, x0 = np.floor(iy), np.floor(ix)
    ay, ax = (iy - y0).astype(np.float32), (ix - x0).astype(np.float32)
    y0, x0 = y0.astype(np.int64), x0.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return np.where(ok, img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], np.float32(0))

    one = np.float32(1)
    return (tap(y0, x0) * ((one - ax) * (one - ay)) + tap(y0, x0 + 1) * (ax * (one - ay))
            + tap(y0 + 1, x0) * ((one - ax) * ay) + tap(y0 + 1, x0 + 1) * (ax * ay))


def ew_reference(info, bufs):
    ip, ib = list(info.ipar), list(info.ibuf)
    k = info.ew
    if k == EW_PP_MASK_F32:
        n = ip[0]
        bufs[ib[1]][:n] = (bufs[ib[0]][:n] != 0).astype(np.float32)
    elif k == EW_PP_IMGPROP:
        Cc, h, w, first, idx, prev, fi, direction = ip[:8]
        hw = h * w
        fe = Cc * hw
        cur = bufs[ib[0]][idx * fe:(idx + 1) * fe].reshape(Cc, h, w)
        mcur = bufs[ib[1]][idx * hw:(idx + 1) * hw].reshape(h, w)
        if first:
            bufs[ib[2]][idx * fe:(idx + 1) * fe] = cur.reshape(-1)
            bufs[ib[3]][idx * hw:(idx + 1) * hw] = mcur.reshape(-1)
            return
        pprop = bufs[ib[2]][prev * fe:(prev + 1) * fe].reshape(Cc, h, w)
        pmask = bufs[ib[3]][prev * hw:(prev + 1) * hw].reshape(h, w)
        fprop = bufs[PB_IN_FLOW_F if direction == 0 else PB_IN_FLOW_B][fi * 2 * hw:(fi + 1) * 2 * hw].reshape(2, h, w)
        fcheck = bufs[PB_IN_FLOW_B if direction == 0 else PB_IN_FLOW_F][fi * 2 * hw:(fi + 1) * 2 * hw].reshape(2, h, w)
        ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
        ix, iy = _warp_coord(xs + fprop[0], w), _warp_coord(ys + fprop[1], h)
        bx, by = _bilinear(fcheck[0], iy, ix), _bilinear(fcheck[1], iy, ix)
        dx, dy = fprop[0] + bx, fprop[1] + by
        diff = dx * dx + dy * dy
        mag = (fprop[0] * fprop[0] + fprop[1] * fprop[1]) + (bx * bx + by * by)
        valid = (diff < np.float32(0.01) * mag + np.float32(0.5)).astype(np.float32)
        mvalid = (_bilinear(pmask, iy, ix) > np.float32(0.1)).astype(np.float32)
        uni = (mcur * valid * (1 - mvalid) > np.float32(0.1)).astype(np.float32)
        nx, ny = np.rint(ix).astype(np.int64), np.rint(iy).astype(np.int64)
        inb = (nx >= 0) & (nx < w) & (ny >= 0) & (ny < h)
        wv = np.where(inb[None], pprop[:, np.clip(ny, 0, h - 1), np.clip(nx, 0, w - 1)], np.float32(0))
        bufs[ib[2]][idx * fe:(idx + 1) * fe] = (uni[None] * wv + (1 - uni[None]) * cur).astype(np.float32).reshape(-1)
        bufs[ib[3]][idx * hw:(idx + 1) * hw] = (mcur * (1 - (valid * (1 - mvalid))) > np.float32(0.1)).astype(np.float32).reshape(-1)
    else:
        raise AssertionError(f"unknown ProPainter op {k}")


def _make_bufs(view, weights=None):
    bufs = []
    for b, n in enumerate(view.buf_elems):
        if b == 0:
            bufs.append(np.asarray(weights if weights is not None else np.zeros(0), dtype=np.float32))
        elif b in BYTE_BUFS:
            bufs.append(np.zeros(n, dtype=np.uint8))
        else:
            bufs.append(np.zeros(n, dtype=np.float32))
    return bufs


def replay_imgprop(view, masked_frames, flows_f, flows_b, masks_u8):
    """masked_frames [t,3,H,W] fp32, flows [t-1,2,H,W], masks_u8 [t,H,W] -> (propagated frames [t,3,H,W], updated masks [t,H,W] u8)"""
    t, _, H, W = masked_frames.shape
    bufs = _make_bufs(view)
    bufs[PB_IN_FRAMES][: masked_frames.size] = masked_frames.reshape(-1)
    bufs[PB_IN_MASK_U8][: masks_u8.size] = masks_u8.reshape(-1)
    if t > 1:
        bufs[PB_IN_FLOW_F][: flows_f.size] = flows_f.reshape(-1)
        bufs[PB_IN_FLOW_B][: flows_b.size] = flows_b.reshape(-1)
    for info, _ in view.ops:
        assert info.kind == OP_EW
        ew_reference(info, bufs)
    return bufs[PB_FW][: t * 3 * H * W].reshape(t, 3, H, W).copy(), (bufs[PB_FWM][: t * H * W].reshape(t, H, W) > 0.5).astype(np.uint8)
