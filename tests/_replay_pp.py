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


# ------------------------------------------------------------------------------------------------
# InpaintGenerator.forward plan (csrc/pp_plan.cpp PpGenPlan, kernels csrc/pp_gen_kernels.hip)
# ------------------------------------------------------------------------------------------------
(EW_PP_IM2COL3, EW_PP_DS_FLOW, EW_PP_DS_MASK, EW_PP_FEATPROP_PREP, EW_PP_DEFORM_COLS, EW_PP_LAYERNORM, EW_PP_POOL, EW_PP_FOLD,
 EW_PP_UNFOLD_GELU, EW_PP_TANH_OUT) = range(33, 43)
PG_OUT = 47


def gen_plan_view(_lib, engine, t, lt, H, W, flags, box=None, mode=0):
    """box = (row_lo, row_hi, col_lo, col_hi): the plan of vsr_pp_forward_box; mode 1 / 2: vsr_pp_encode / vsr_pp_forward_cached"""
    p = C.c_void_p()
    f = np.ascontiguousarray(flags if flags is not None else np.zeros(0), dtype=np.uint8)
    if mode:
        _lib.check(_lib.lib.vsr_pp_gen_plan_create_mode(engine.handle, t, lt, H, W, f.ctypes.data_as(C.c_void_p) if f.size else None, f.size,
                                                        *[int(b) for b in (box or (0, 0, 0, 0))], mode, C.byref(p)))
    elif box is not None:
        _lib.check(_lib.lib.vsr_pp_gen_plan_create_box(engine.handle, t, lt, H, W, f.ctypes.data_as(C.c_void_p), f.size, *[int(b) for b in box],
                                                       C.byref(p)))
    else:
        _lib.check(_lib.lib.vsr_pp_gen_plan_create(engine.handle, t, lt, H, W, f.ctypes.data_as(C.c_void_p), f.size, C.byref(p)))
    return _replay.PlanView(_lib, None, 0, plan_ptr=p)


def _sig(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float32)))).astype(np.float32)


def _slot(buf, off, h, w, halo, Cc):
    Hp, Wp = h + 2 * halo, w + 2 * halo
    full = buf[off: off + Hp * Wp * Cc].reshape(Hp, Wp, Cc)
    return full[halo:halo + h, halo:halo + w, :]


def gen_ew_reference(info, bufs):
    ip, ib, io = list(info.ipar), list(info.ibuf), list(info.ioff)
    k = info.ew
    if k == EW_PP_IM2COL3:
        n, H, W = ip[:3]
        fr = bufs[ib[0]][: n * 3 * H * W].reshape(n, 3, H, W)
        m1 = (bufs[ib[1]][: n * H * W].reshape(n, 1, H, W) != 0).astype(np.float32)
        m2 = (bufs[ib[2]][: n * H * W].reshape(n, 1, H, W) != 0).astype(np.float32)
        x = torch.from_numpy(np.concatenate([fr, m1, m2], 1))
        cols = torch.nn.functional.unfold(x, kernel_size=3, padding=1, stride=2)       # n, (c,ky,kx), oh*ow
        oh, ow = H // 2, W // 2
        cols = cols.view(n, 5, 9, oh * ow).permute(0, 3, 2, 1).reshape(n * oh * ow, 45)
        out = np.zeros((n * oh * ow, 64), dtype=np.float32)
        out[:, :45] = cols.numpy()
        bufs[ib[3]][: out.size] = out.reshape(-1)
    elif k == EW_PP_DS_FLOW:
        n2, H, W = ip[:3]
        s = bufs[ib[0]][: n2 * H * W].reshape(n2, H, W)
        half = np.float32(0.5)
        top = half * s[:, 1::4, 1::4] + half * s[:, 1::4, 2::4]
        bot = half * s[:, 2::4, 1::4] + half * s[:, 2::4, 2::4]
        bufs[ib[1]][: n2 * (H // 4) * (W // 4)] = ((half * top + half * bot) / np.float32(4.0)).reshape(-1)
    elif k == EW_PP_DS_MASK:
        n, H, W, halo, Cc = ip[:5]
        h, w = H // 4, W // 4
        fe = (h + 2 * halo) * (w + 2 * halo) * Cc
        m1 = (bufs[ib[0]][: n * H * W].reshape(n, H, W)[:, ::4, ::4] != 0).astype(np.float32)
        m2 = (bufs[ib[1]][: n * H * W].reshape(n, H, W)[:, ::4, ::4] != 0).astype(np.float32)
        for f in range(n):
            s = _slot(bufs[ib[2]], io[2] + f * fe, h, w, halo, Cc)
            s[:, :, 0] = m1[f]
            s[:, :, 1] = m2[f]
    elif k == EW_PP_FEATPROP_PREP:
        h, w, halo, Cc, s_warp, s_misc = ip[:6]
        fe = (h + 2 * halo) * (w + 2 * halo) * Cc
        prop = _slot(bufs[ib[0]], io[0], h, w, halo, Cc)
        fprop = bufs[ib[1]][io[1]: io[1] + 2 * h * w].reshape(2, h, w)
        fcheck = bufs[ib[2]][io[2]: io[2] + 2 * h * w].reshape(2, h, w)
        mk = _slot(bufs[ib[0]], io[3], h, w, halo, Cc)
        ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
        ix, iy = _warp_coord(xs + fprop[0], w), _warp_coord(ys + fprop[1], h)
        y0, x0 = np.floor(iy), np.floor(ix)
        ay, ax = (iy - y0).astype(np.float32), (ix - x0).astype(np.float32)
        y0, x0 = y0.astype(np.int64), x0.astype(np.int64)
        acc = np.zeros((h, w, Cc), dtype=np.float32)
        one = np.float32(1)
        for dy, dx, wgt in ((0, 0, (one - ax) * (one - ay)), (0, 1, ax * (one - ay)), (1, 0, (one - ax) * ay), (1, 1, ax * ay)):
            yy, xx = y0 + dy, x0 + dx
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            acc += prop[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)] * (wgt * ok)[..., None].astype(np.float32)
        _slot(bufs[ib[0]], s_warp * fe, h, w, halo, Cc)[...] = acc
        bx, by = _bilinear(fcheck[0], iy, ix), _bilinear(fcheck[1], iy, ix)
        dx, dy = fprop[0] + bx, fprop[1] + by
        valid = (dx * dx + dy * dy < np.float32(0.01) * ((fprop[0] ** 2 + fprop[1] ** 2) + (bx * bx + by * by)) + np.float32(0.5))
        misc = _slot(bufs[ib[0]], s_misc * fe, h, w, halo, Cc)
        misc[:, :, 0], misc[:, :, 1], misc[:, :, 2] = fprop[0], fprop[1], valid.astype(np.float32)
        misc[:, :, 3], misc[:, :, 4] = mk[:, :, 0], mk[:, :, 1]
    elif k == EW_PP_DEFORM_COLS:
        h, w, halo, Cc, ld = ip[:5]
        mag = np.float32(info.fpar[0])
        x = _slot(bufs[ib[0]], io[0], h, w, halo, Cc)
        o = bufs[ib[1]][: h * w * ld].reshape(h, w, ld)
        fl = bufs[ib[2]][io[2]: io[2] + 2 * h * w].reshape(2, h, w)
        offs = mag * np.tanh(o[..., :288]).reshape(h, w, 16, 9, 2)
        msk = _sig(o[..., 288:432]).reshape(h, w, 16, 9)
        ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
        cols = np.zeros((h, w, Cc // 32, 9, 32), dtype=np.float32)
        xg = x.reshape(h * w, 16, Cc // 16)
        for g in range(16):
            for kk in range(9):
                py = ys - 1 + kk // 3 + (offs[..., g, kk, 0] + fl[1])
                px = xs - 1 + kk % 3 + (offs[..., g, kk, 1] + fl[0])
                y0, x0 = np.floor(py), np.floor(px)
                ly, lx = (py - y0).astype(np.float32), (px - x0).astype(np.float32)
                y0, x0 = y0.astype(np.int64), x0.astype(np.int64)
                acc = np.zeros((h, w, Cc // 16), dtype=np.float32)
                for dy, dx, wgt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx), (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
                    yy, xx = y0 + dy, x0 + dx
                    ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                    acc += xg[np.clip(yy, 0, h - 1) * w + np.clip(xx, 0, w - 1), g] * (wgt * ok)[..., None].astype(np.float32)
                ci0 = g * (Cc // 16)
                cols[:, :, ci0 // 32, kk, ci0 % 32: ci0 % 32 + Cc // 16] = acc * msk[..., g, kk][..., None]
        bufs[ib[3]][: cols.size] = cols.reshape(-1)
    elif k == EW_PP_LAYERNORM:
        t, fh, fw, Cc, gh, gw = ip[:6]
        W_ = bufs[0]
        x = torch.from_numpy(bufs[ib[0]][: t * fh * fw * Cc].reshape(t, fh, fw, Cc).copy())
        y = torch.nn.functional.layer_norm(x, (Cc,), torch.from_numpy(W_[io[0]: io[0] + Cc].copy()), torch.from_numpy(W_[io[1]: io[1] + Cc].copy()))
        bufs[ib[1]][: t * gh * gw * Cc].reshape(t, gh, gw, Cc)[:, :fh, :fw, :] = y.numpy()
    elif k == EW_PP_POOL:
        t, gh, gw, Cc, ph, pw = ip[:6]
        W_ = bufs[0]
        y = torch.from_numpy(bufs[ib[0]][: t * gh * gw * Cc].reshape(t, gh, gw, Cc).copy()).permute(0, 3, 1, 2)
        wt = torch.from_numpy(W_[io[2]: io[2] + Cc * 16].reshape(Cc, 1, 4, 4).copy())
        bs = torch.from_numpy(W_[io[3]: io[3] + Cc].copy())
        out = torch.nn.functional.conv2d(y, wt, bs, stride=4, groups=Cc).permute(0, 2, 3, 1)
        assert out.shape[1:3] == (ph, pw)
        bufs[ib[0]][io[1]: io[1] + out.numel()] = out.reshape(-1).numpy()
    elif k == EW_PP_FOLD:
        ld, t, fh, fw, h, w, Cc, halo, normalize = ip[:9]
        # the token vectors are tap-major (element tap*C + c: csrc/pp_gen_kernels.hip); F.fold wants c*49 + tap
        v = bufs[ib[0]][: t * fh * fw * ld].reshape(t, fh * fw, ld)[:, :, :Cc * 49].reshape(t, fh * fw, 49, Cc).transpose(0, 1, 3, 2)
        v = torch.from_numpy(np.ascontiguousarray(v).reshape(t, fh * fw, Cc * 49)).permute(0, 2, 1)
        y = torch.nn.functional.fold(v, (h, w), (7, 7), padding=(3, 3), stride=(3, 3))
        if normalize:
            y = y / torch.nn.functional.fold(torch.ones_like(v), (h, w), (7, 7), padding=(3, 3), stride=(3, 3))
        Hp, Wp = h + 2 * halo, w + 2 * halo
        bufs[ib[1]][: t * Hp * Wp * Cc].reshape(t, Hp, Wp, Cc)[:, halo:halo + h, halo:halo + w, :] = y.permute(0, 2, 3, 1).numpy()
    elif k == EW_PP_UNFOLD_GELU:
        t, fh, fw, h, w, Cc, ld = ip[:7]
        m = torch.from_numpy(bufs[ib[0]][: t * h * w * Cc].reshape(t, h, w, Cc).copy()).permute(0, 3, 1, 2)
        u = torch.nn.functional.unfold(m, (7, 7), padding=(3, 3), stride=(3, 3)).permute(0, 2, 1)            # t, tokens, Cc*49
        out = np.zeros((t * fh * fw, ld), dtype=np.float32)
        g = torch.nn.functional.gelu(u).reshape(t * fh * fw, Cc, 49).numpy()                             # F.unfold: c*49 + tap
        out[:, :Cc * 49] = g.transpose(0, 2, 1).reshape(t * fh * fw, Cc * 49)                              # stored tap-major
        bufs[ib[1]][: out.size] = out.reshape(-1)
    elif k == EW_PP_TANH_OUT:
        ld, n, H, W = ip[:4]
        y = bufs[ib[0]][: n * H * W * ld].reshape(n, H, W, ld)[..., :3]
        bufs[ib[1]][: n * 3 * H * W] = np.tanh(y).transpose(0, 3, 1, 2).reshape(-1)
    else:
        raise AssertionError(f"unknown generator op {k}")


PG_FEAT, PG_PROP, PG_X, PG_TOKOUT = 17, 20, 28, 48         # csrc/pp_plan.h PpBuf


def replay_encode(view, packed_weights, frames, masks_in_u8, masks_upd_u8):
    """PP_PLAN_ENCODE: -> (features [n,h,w,128], tokens [n, fh*fw, 512]) as vsr_pp_encode returns them"""
    n, _, H, W = frames.shape
    h, w = H // 4, W // 4
    out, bufs = replay_gen(view, packed_weights, frames, None, None, masks_in_u8, masks_upd_u8, 1, want_out=False)
    feat = bufs[PG_FEAT][: n * (h + 6) * (w + 6) * 128].reshape(n, h + 6, w + 6, 128)[:, 3:3 + h, 3:3 + w, :].copy()
    return feat, bufs[PG_TOKOUT].reshape(-1, (((h - 1) // 3) + 1) * (((w - 1) // 3) + 1), 512).copy()       # tokens of the first lt frames


def replay_gen(view, packed_weights, frames, flows_f, flows_b, masks_in_u8, masks_upd_u8, lt, want_out=True, cached=None):
    """frames [t,3,H,W] fp32, flows [lt-1,2,H,W], masks u8 [t,H,W] -> (tanh output [lt,3,H,W], buffers).
    cached = (features [t,h,w,128], tokens [t,ntok,512]) of the window's frames (PP_PLAN_CACHED): what vsr_pp_forward_cached copies in --
    the local frames' features into the propagation buffer's input slots, the reference frames' tokens into the token buffer."""
    t, _, H, W = frames.shape
    bufs = _make_bufs(view, packed_weights)
    if cached is not None:
        feats, toks = cached
        h, w = H // 4, W // 4
        slots = bufs[PG_PROP].reshape(-1, h + 2, w + 2, 128)
        slots[:lt, 1:1 + h, 1:1 + w, :] = feats[:lt]
        ntok = toks.shape[1]
        bufs[PG_X][lt * ntok * 512: t * ntok * 512] = toks[lt:].reshape(-1)
    else:
        bufs[PB_IN_FRAMES][: frames.size] = frames.reshape(-1)
    if not want_out:
        lt = 0
    bufs[PB_IN_MASK_U8][: masks_in_u8.size] = masks_in_u8.reshape(-1)
    bufs[PB_IN_MASK_UPD_U8][: masks_upd_u8.size] = masks_upd_u8.reshape(-1)
    if lt > 1:
        bufs[PB_IN_FLOW_F][: flows_f.size] = flows_f.reshape(-1)
        bufs[PB_IN_FLOW_B][: flows_b.size] = flows_b.reshape(-1)
    with torch.no_grad():
        for info, items in view.ops:
            if info.kind == _replay.OP_GEMM:
                for it in items:
                    _replay.gemm_reference(it, info.bmode, bufs, view.tables)
            elif info.kind == _replay.OP_SOFTMAX:
                for it in items:
                    _replay.softmax_reference(it, bufs)
            elif info.kind == _replay.OP_UPSAMPLE2X:
                _replay.upsample_reference(info, bufs)
            elif info.kind == OP_EW:
                gen_ew_reference(info, bufs)
            else:
                raise AssertionError(f"unexpected op kind {info.kind}")
    if not want_out:
        return None, bufs
    return bufs[PG_OUT][: lt * 3 * H * W].reshape(lt, 3, H, W).copy(), bufs
