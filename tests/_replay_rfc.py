"""CPU replay of the flow-completion plan (test infrastructure): OP_EW sub-kinds of csrc/rfc_plan.h executed with numpy /
torch-CPU exactly as csrc/flow_kernels.hip defines them; GEMMs and the 2x upsampling through tests/_replay."""
import ctypes as C

import numpy as np
import torch

import _replay

EW_RFC_IM2COL5, EW_DEFORM_COLS, EW_RFC_COMBINE = 20, 21, 22
FB_WEIGHTS, FB_IN_FLOW_F, FB_IN_FLOW_B, FB_IN_MASK = 0, 1, 2, 3
FB_OUT_F, FB_OUT_B = 33, 34
OP_EW = 6


def rfc_plan_view(_lib, engine, t, H, W):
    p = C.c_void_p()
    _lib.check(_lib.lib.vsr_rfc_plan_create(engine.handle, t, H, W, C.byref(p)))
    return _replay.PlanView(_lib, None, 0, plan_ptr=p)


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float32)))).astype(np.float32)


def ew_reference(info, bufs):
    ip, ib, io = list(info.ipar), list(info.ibuf), list(info.ioff)
    k = info.ew
    if k == EW_RFC_IM2COL5:
        t, H, W = ip[:3]
        T = t - 1
        ff = bufs[ib[0]][: T * 2 * H * W].reshape(T, 2, H, W)
        fb = bufs[ib[1]][: T * 2 * H * W].reshape(T, 2, H, W)
        mk = (bufs[ib[2]][: t * H * W].reshape(t, 1, H, W) != 0).astype(np.float32)
        seq0 = np.concatenate([ff * (1 - mk[:-1]), mk[:-1]], 1)                       # [T,3,H,W]
        seq1 = np.concatenate([fb * (1 - mk[1:]), mk[1:]], 1)[::-1]                   # flipped in time
        x = np.stack([seq0, seq1], 1).reshape(2 * T, 3, H, W)                         # frame = step*2 + s
        xt = torch.nn.functional.pad(torch.from_numpy(np.ascontiguousarray(x)), (2, 2, 2, 2), mode="replicate")
        cols = torch.nn.functional.unfold(xt, kernel_size=5, stride=2)                # n, (c,ky,kx), oh*ow
        oh, ow = H // 2, W // 2
        cols = cols.view(2 * T, 3, 25, oh * ow).permute(0, 3, 2, 1).reshape(2 * T * oh * ow, 75)
        out = np.zeros((2 * T * oh * ow, 96), dtype=np.float32)
        out[:, :75] = cols.numpy()
        bufs[ib[3]][: out.size] = out.reshape(-1)
    elif k == EW_DEFORM_COLS:
        n, h, w, halo, Cc, ld = ip[:6]
        mag = np.float32(info.fpar[0])
        Hp, Wp = h + 2 * halo, w + 2 * halo
        fe = Hp * Wp * Cc

        def interior(off):
            return bufs[ib[0]][off: off + n * fe].reshape(n, Hp, Wp, Cc)[:, halo:halo + h, halo:halo + w, :]

        x = np.concatenate([interior(io[0]), interior(io[1])], -1)                    # [n,h,w,256]
        o = bufs[ib[1]][: n * h * w * ld].reshape(n, h, w, ld)
        offs = mag * np.tanh(o[..., :288]).reshape(n, h, w, 16, 9, 2)
        msk = _sigmoid(o[..., 288:432]).reshape(n, h, w, 16, 9)
        ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
        cols = np.zeros((n, h, w, 8, 9, 32), dtype=np.float32)                        # [.., ci/32, tap, ci%32]
        xg = x.reshape(n, h * w, 16, 16)
        fi = np.arange(n)[:, None, None]
        for g in range(16):
            for kk in range(9):
                py = ys[None] - 1 + kk // 3 + offs[..., g, kk, 0]
                px = xs[None] - 1 + kk % 3 + offs[..., g, kk, 1]
                y0, x0 = np.floor(py), np.floor(px)
                ly, lx = (py - y0).astype(np.float32), (px - x0).astype(np.float32)
                y0, x0 = y0.astype(np.int64), x0.astype(np.int64)
                acc = np.zeros((n, h, w, 16), dtype=np.float32)
                for dy, dx, wgt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx), (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
                    yy, xx = y0 + dy, x0 + dx
                    ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                    v = xg[fi, np.clip(yy, 0, h - 1) * w + np.clip(xx, 0, w - 1), g]        # [n,h,w,16]
                    acc += v * (wgt * ok)[..., None].astype(np.float32)
                ci0 = g * 16
                cols[:, :, :, ci0 // 32, kk, ci0 % 32: ci0 % 32 + 16] = acc * msk[..., g, kk][..., None]
        bufs[ib[2]][: cols.size] = cols.reshape(-1)
    elif k == EW_RFC_COMBINE:
        t, H, W, ld = ip[:4]
        T = t - 1
        pred = bufs[ib[0]][: 2 * T * H * W * ld].reshape(T, 2, H, W, ld)[..., :2]       # [step, s, H, W, 2]
        ff = bufs[FB_IN_FLOW_F][: T * 2 * H * W].reshape(T, 2, H, W)
        fb = bufs[FB_IN_FLOW_B][: T * 2 * H * W].reshape(T, 2, H, W)
        mk = (bufs[FB_IN_MASK][: t * H * W].reshape(t, 1, H, W) != 0).astype(np.float32)
        pf = pred[:, 0].transpose(0, 3, 1, 2)
        pb = pred[::-1, 1].transpose(0, 3, 1, 2)                                          # un-flip
        bufs[ib[1]][: T * 2 * H * W] = (pf * mk[:-1] + ff * (1 - mk[:-1]) * (1 - mk[:-1])).astype(np.float32).reshape(-1)
        bufs[ib[2]][: T * 2 * H * W] = (pb * mk[1:] + fb * (1 - mk[1:]) * (1 - mk[1:])).astype(np.float32).reshape(-1)
    else:
        raise AssertionError(f"unknown flow-completion op {k}")


def replay_rfc(view, packed_weights, flows_f, flows_b, masks_u8):
    """flows [t-1,2,H,W] fp32, masks_u8 [t,H,W] -> (completed forward, completed backward), buffers."""
    T, _, H, W = flows_f.shape
    bufs = []
    for b, n in enumerate(view.buf_elems):
        if b == FB_WEIGHTS:
            bufs.append(np.asarray(packed_weights, dtype=np.float32))
        elif b == FB_IN_MASK:
            a = np.zeros(n, dtype=np.uint8)
            a[: masks_u8.size] = masks_u8.reshape(-1)
            bufs.append(a)
        else:
            bufs.append(np.zeros(n, dtype=np.float32))
    bufs[FB_IN_FLOW_F][: flows_f.size] = flows_f.reshape(-1)
    bufs[FB_IN_FLOW_B][: flows_b.size] = flows_b.reshape(-1)
    with torch.no_grad():
        for info, items in view.ops:
            if info.kind == _replay.OP_GEMM:
                for it in items:
                    _replay.gemm_reference(it, info.bmode, bufs, view.tables)
            elif info.kind == _replay.OP_UPSAMPLE2X:
                _replay.upsample_reference(info, bufs)
            elif info.kind == OP_EW:
                ew_reference(info, bufs)
            else:
                raise AssertionError(f"unexpected op kind {info.kind} in a flow-completion plan")
    n = T * 2 * H * W
    return bufs[FB_OUT_F][:n].reshape(T, 2, H, W).copy(), bufs[FB_OUT_B][:n].reshape(T, 2, H, W).copy(), bufs
