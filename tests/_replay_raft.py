"""CPU replay of the RAFT plan (test infrastructure): the OP_EW sub-kinds of csrc/raft_plan.h executed with numpy /
torch-CPU exactly as csrc/flow_kernels.hip defines them, GEMMs through tests/_replay.gemm_reference."""
import ctypes as C

import numpy as np
import torch

import _replay

(EW_IM2COL7_U8, EW_INORM_STATS, EW_INORM_APPLY, EW_CTX_SPLIT, EW_FLOW_UPDATE, EW_IM2COL7_FLOW, EW_AVGPOOL2, EW_CORR_LOOKUP, EW_GRU_RH,
 EW_GRU_UPDATE, EW_CONVEX_UP, EW_CORR_TRANSPOSE) = range(1, 13)
RB_WEIGHTS, RB_IN_U8, RB_OUT = 0, 1, 30
OP_EW = 6


def raft_plan_view(_lib, engine, t, H, W, iters):
    p = C.c_void_p()
    _lib.check(_lib.lib.vsr_raft_plan_create(engine.handle, t, H, W, iters, C.byref(p)))
    return _replay.PlanView(_lib, None, 0, plan_ptr=p)


def _nhwc(buf, n, H, W, Cc, halo):
    """(padded view [n,Hp,Wp,C], interior view) of an activation buffer."""
    Hp, Wp = H + 2 * halo, W + 2 * halo
    full = buf[: n * Hp * Wp * Cc].reshape(n, Hp, Wp, Cc)
    return full, full[:, halo:halo + H, halo:halo + W, :]


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float32)))).astype(np.float32)


def ew_reference(info, bufs, tables):
    ip, ib, io = list(info.ipar), list(info.ibuf), list(info.ioff)
    k = info.ew
    if k == EW_IM2COL7_U8:
        n, H, W = ip[0], ip[1], ip[2]
        img = bufs[ib[0]][: n * H * W * 3].reshape(n, H, W, 3)
        x = torch.from_numpy(np.ascontiguousarray(img)).permute(0, 3, 1, 2).float().div(255) * 2 - 1
        cols = torch.nn.functional.unfold(x, kernel_size=7, padding=3, stride=2)            # n, (c,ky,kx), oh*ow
        oh, ow = H // 2, W // 2
        cols = cols.view(n, 3, 49, oh * ow).permute(0, 3, 2, 1).reshape(n * oh * ow, 147)    # k = tap*3 + c
        out = np.zeros((n * oh * ow, 160), dtype=np.float32)
        out[:, :147] = cols.numpy()
        bufs[ib[1]][: out.size] = out.reshape(-1)
    elif k == EW_INORM_STATS:
        n, H, W, Cc, halo = ip[:5]
        _, x = _nhwc(bufs[ib[0]], n, H, W, Cc, halo)
        x64 = x.astype(np.float64)
        mean = x64.mean(axis=(1, 2))
        var = (x64 * x64).mean(axis=(1, 2)) - mean * mean
        st = np.stack([mean, 1.0 / np.sqrt(np.maximum(var, 0) + 1e-5)], axis=-1).astype(np.float32)      # [n,C,2]
        bufs[ib[1]][: st.size] = st.reshape(-1)
    elif k == EW_INORM_APPLY:
        n, H, W, Cc, halo, relu, res_halo = ip[:7]
        _, x = _nhwc(bufs[ib[0]], n, H, W, Cc, halo)
        st = bufs[ib[1]][: n * Cc * 2].reshape(n, 1, 1, Cc, 2)
        y = (x - st[..., 0]) * st[..., 1]
        if relu:
            y = np.maximum(y, 0)
        if ib[2] >= 0:
            _, r = _nhwc(bufs[ib[2]], n, H, W, Cc, res_halo)
            y = np.maximum(y + r, 0)
        x[...] = y.astype(np.float32)
    elif k == EW_CTX_SPLIT:
        pairs, h, w, halo, chx, tid = ip[:6]
        cm = bufs[ib[0]]
        _, hx = _nhwc(bufs[ib[1]], pairs, h, w, chx, halo)
        for p in range(pairs):
            f = int(tables[tid][p])
            c = cm[f * h * w * 256:(f + 1) * h * w * 256].reshape(h, w, 256)
            hx[p, :, :, :128] = np.tanh(c[:, :, :128])
            hx[p, :, :, 128:256] = np.maximum(c[:, :, 128:], 0)
    elif k == EW_FLOW_UPDATE:
        pairs, h, w, init, halo, chx, chflow, ld = ip[:8]
        M = pairs * h * w
        coords = bufs[ib[1]][: M * 2].reshape(pairs, h, w, 2)
        ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
        c0 = np.stack([xs, ys], -1)[None]
        if init:
            coords[...] = c0
        else:
            d = bufs[ib[0]][: M * ld].reshape(pairs, h, w, ld)[..., :2]
            coords[...] = coords + d
        fl = (coords - c0).astype(np.float32)
        bufs[ib[2]][: M * 2] = fl.reshape(-1)
        _, hx = _nhwc(bufs[ib[3]], pairs, h, w, chx, halo)
        hx[..., chflow:chflow + 2] = fl
    elif k == EW_IM2COL7_FLOW:
        pairs, h, w = ip[:3]
        fl = bufs[ib[0]][: pairs * h * w * 2].reshape(pairs, h, w, 2)
        x = torch.from_numpy(np.ascontiguousarray(fl)).permute(0, 3, 1, 2)
        cols = torch.nn.functional.unfold(x, kernel_size=7, padding=3)                      # n, (c,ky,kx), h*w
        cols = cols.view(pairs, 2, 49, h * w).permute(0, 3, 2, 1).reshape(pairs * h * w, 98)
        out = np.zeros((pairs * h * w, 128), dtype=np.float32)
        out[:, :98] = cols.numpy()
        bufs[ib[1]][: out.size] = out.reshape(-1)
    elif k == EW_CORR_TRANSPOSE:
        n, hw = ip[:2]
        src = bufs[ib[0]][io[0]: io[0] + n * hw * hw].reshape(n, hw, hw)
        bufs[ib[0]][io[1]: io[1] + n * hw * hw] = np.ascontiguousarray(src.transpose(0, 2, 1)).reshape(-1)
    elif k == EW_AVGPOOL2:
        rows, hs, ws = ip[:3]
        src = bufs[ib[0]][io[0]: io[0] + rows * hs * ws].reshape(rows, hs, ws)
        hd, wd = hs // 2, ws // 2
        s = src[:, :2 * hd, :2 * wd]
        out = (s[:, 0::2, 0::2] + s[:, 0::2, 1::2] + s[:, 1::2, 0::2] + s[:, 1::2, 1::2]) * np.float32(0.25)
        bufs[ib[0]][io[1]: io[1] + rows * hd * wd] = out.reshape(-1)
    elif k == EW_CORR_LOOKUP:
        M, ld = ip[0], ip[9]
        coords = bufs[ib[1]][: M * 2].reshape(M, 2)
        out = np.zeros((M, ld), dtype=np.float32)
        offs = np.arange(-4, 5, dtype=np.float32)
        rows = np.arange(M)[:, None, None]
        for lvl in range(4):
            hh, ww = ip[1 + lvl], ip[5 + lvl]
            vol = bufs[ib[0]][io[lvl]: io[lvl] + M * hh * ww].reshape(M, hh, ww)
            sc = np.float32(2 ** lvl)
            x = (coords[:, 0] / sc)[:, None, None] + offs[None, :, None]          # window index i walks x
            y = (coords[:, 1] / sc)[:, None, None] + offs[None, None, :]          # j walks y
            x, y = np.broadcast_arrays(x, y)
            gx = np.float32(2.0) * x / np.float32(ww - 1) - np.float32(1.0)
            gy = np.float32(2.0) * y / np.float32(hh - 1) - np.float32(1.0)
            ix = ((gx + np.float32(1.0)) / np.float32(2.0)) * np.float32(ww - 1)
            iy = ((gy + np.float32(1.0)) / np.float32(2.0)) * np.float32(hh - 1)
            x0, y0 = np.floor(ix), np.floor(iy)
            ax, ay = (ix - x0).astype(np.float32), (iy - y0).astype(np.float32)
            x0, y0 = x0.astype(np.int64), y0.astype(np.int64)

            def tap(yy, xx):
                ok = (yy >= 0) & (yy < hh) & (xx >= 0) & (xx < ww)
                v = vol[rows, np.clip(yy, 0, hh - 1), np.clip(xx, 0, ww - 1)]
                return np.where(ok, v, np.float32(0))

            one = np.float32(1.0)
            val = (tap(y0, x0) * ((one - ax) * (one - ay)) + tap(y0, x0 + 1) * (ax * (one - ay))
                   + tap(y0 + 1, x0) * ((one - ax) * ay) + tap(y0 + 1, x0 + 1) * (ax * ay))
            out[:, lvl * 81:(lvl + 1) * 81] = val.reshape(M, 81)
        bufs[ib[2]][: out.size] = out.reshape(-1)
    elif k == EW_GRU_RH:
        pairs, h, w, halo, chx, ch_h, ch_rh = ip[:7]
        M = pairs * h * w
        zr = bufs[ib[0]][: M * 256].reshape(pairs, h, w, 256)
        _, hx = _nhwc(bufs[ib[1]], pairs, h, w, chx, halo)
        hx[..., ch_rh:ch_rh + 128] = _sigmoid(zr[..., 128:]) * hx[..., ch_h:ch_h + 128]
    elif k == EW_GRU_UPDATE:
        pairs, h, w, halo, chx, ch_h = ip[:6]
        M = pairs * h * w
        z = _sigmoid(bufs[ib[0]][: M * 256].reshape(pairs, h, w, 256)[..., :128])
        q = np.tanh(bufs[ib[1]][: M * 128].reshape(pairs, h, w, 128))
        _, hx = _nhwc(bufs[ib[2]], pairs, h, w, chx, halo)
        hx[..., ch_h:ch_h + 128] = (np.float32(1.0) - z) * hx[..., ch_h:ch_h + 128] + z * q
    elif k == EW_CONVEX_UP:
        pairs, h, w = ip[:3]
        M = pairs * h * w
        fl = torch.from_numpy(bufs[ib[0]][: M * 2].reshape(pairs, h, w, 2).copy()).permute(0, 3, 1, 2)
        mk = torch.from_numpy(bufs[ib[1]][: M * 576].reshape(pairs, h, w, 576).copy()).permute(0, 3, 1, 2)
        m = torch.softmax(mk.reshape(pairs, 1, 9, 8, 8, h, w), dim=2)
        nb = torch.nn.functional.unfold(8 * fl, [3, 3], padding=1).reshape(pairs, 2, 9, 1, 1, h, w)
        up = torch.sum(m * nb, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(pairs, 2, 8 * h, 8 * w)
        bufs[ib[2]][: up.numel()] = up.numpy().reshape(-1)
    else:
        raise AssertionError(f"unknown RAFT op {k}")


def replay_raft(view, packed_weights, frames_u8):
    """frames_u8 [t,H,W,3] RGB -> (forward flows, backward flows) [t-1,2,H,W], buffers."""
    t, H, W, _ = frames_u8.shape
    bufs = []
    for b, n in enumerate(view.buf_elems):
        if b == RB_WEIGHTS:
            bufs.append(np.asarray(packed_weights, dtype=np.float32))
        elif b == RB_IN_U8:
            a = np.zeros(n, dtype=np.uint8)
            a[: frames_u8.size] = frames_u8.reshape(-1)
            bufs.append(a)
        else:
            bufs.append(np.zeros(n, dtype=np.float32))
    with torch.no_grad():
        for info, items in view.ops:
            if info.kind == _replay.OP_GEMM:
                for it in items:
                    _replay.gemm_reference(it, info.bmode, bufs, view.tables)
            elif info.kind == OP_EW:
                ew_reference(info, bufs, view.tables)
            else:
                raise AssertionError(f"unexpected op kind {info.kind} in a RAFT plan")
    out = bufs[RB_OUT][: 2 * (t - 1) * 2 * H * W].reshape(2, t - 1, 2, H, W)
    return out[0].copy(), out[1].copy(), bufs
