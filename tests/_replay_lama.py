"""CPU replay of the LaMa plan (test infrastructure): the OP_EW sub-kinds of csrc/lama_plan.h executed with numpy exactly as
csrc/lama_kernels.hip defines them, GEMMs (convolutions and DFT stages alike) through tests/_replay.gemm_reference."""
import ctypes as C

import numpy as np
import torch

import _replay

EW_LAMA_IM2COL7, EW_LAMA_HALO, EW_LAMA_ADD_HALO, EW_LAMA_OUT = 50, 51, 52, 53
LB_WEIGHTS, LB_IN_U8, LB_MASK_U8, LB_OUT_U8 = 0, 1, 2, 20
BUF_PLAN_CONST = -2
OP_EW = 6


def lama_plan_view(_lib, engine, B, H, W):
    p = C.c_void_p()
    _lib.check(_lib.lib.vsr_lama_plan_create(engine.handle, B, H, W, C.byref(p)))
    view = _replay.PlanView(_lib, None, 0, plan_ptr=p)
    n = _lib.lib.vsr_plan_consts(p, None, 0)
    view.consts = np.zeros(max(n, 1), dtype=np.float32)
    _lib.lib.vsr_plan_consts(p, view.consts.ctypes.data_as(C.c_void_p), n)
    return view


def _sym(i, n):
    return np.where(i < n, i, 2 * n - 1 - i)


def _reflect(i, n):
    i = np.where(i < 0, -i, i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def ew_reference(info, bufs):
    ip, ib = list(info.ipar), list(info.ibuf)
    k = info.ew
    if k == EW_LAMA_IM2COL7:
        B, H, W, Hp, Wp = ip[:5]
        img = bufs[ib[0]][: B * H * W * 3].reshape(B, H, W, 3)
        msk = bufs[ib[1]][: B * H * W].reshape(B, H, W)
        ys, xs = _sym(np.arange(Hp), H), _sym(np.arange(Wp), W)
        m = (msk[:, ys][:, :, xs] > 0).astype(np.float32)                                  # [B,Hp,Wp]
        x = img[:, ys][:, :, xs].astype(np.float32) / np.float32(255)
        x4 = np.concatenate([x * (np.float32(1) - m)[..., None], m[..., None]], axis=-1)  # [B,Hp,Wp,4]
        cols = np.zeros((B, Hp, Wp, 224), dtype=np.float32)
        for ky in range(7):
            py = _reflect(np.arange(Hp) + ky - 3, Hp)
            for kx in range(7):
                px = _reflect(np.arange(Wp) + kx - 3, Wp)
                tap = ky * 7 + kx
                cols[..., tap * 4: tap * 4 + 4] = x4[:, py][:, :, px]
        bufs[ib[2]][: cols.size] = cols.reshape(-1)
    elif k in (EW_LAMA_HALO, EW_LAMA_ADD_HALO):
        n, H, W, Cc, halo = ip[:5]
        Hp, Wp = H + 2 * halo, W + 2 * halo
        ys, xs = _reflect(np.arange(Hp) - halo, H) + halo, _reflect(np.arange(Wp) - halo, W) + halo
        if k == EW_LAMA_HALO:
            x = bufs[ib[0]][: n * Hp * Wp * Cc].reshape(n, Hp, Wp, Cc)
            x[:] = x[:, ys][:, :, xs]
        else:
            a = bufs[ib[0]][: n * Hp * Wp * Cc].reshape(n, Hp, Wp, Cc)
            b = bufs[ib[1]][: n * Hp * Wp * Cc].reshape(n, Hp, Wp, Cc)
            d = bufs[ib[2]][: n * Hp * Wp * Cc].reshape(n, Hp, Wp, Cc)
            s = a + b
            if ip[5]:
                d[:] = s[:, ys][:, :, xs]
            else:
                d[:, halo:halo + H, halo:halo + W] = s[:, halo:halo + H, halo:halo + W]
    elif k == EW_LAMA_OUT:
        B, H, W, Hp, Wp = ip[:5]
        blk = bufs[ib[0]][: B * (Hp // 4) * (Wp // 4) * 64].reshape(B, Hp // 4, Wp // 4, 64)[..., :48].reshape(B, Hp // 4, Wp // 4, 4, 4, 3)
        lg = blk.transpose(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, 3)[:, :H, :W]       # logits arrive as 4x4 pixel blocks
        img = bufs[ib[1]][: B * H * W * 3].reshape(B, H, W, 3).astype(np.float32) / np.float32(255)
        m = (bufs[ib[2]][: B * H * W].reshape(B, H, W, 1) > 0).astype(np.float32)
        p = torch.sigmoid(torch.from_numpy(np.ascontiguousarray(lg))).numpy()
        v = (m * p + (np.float32(1) - m) * img) * np.float32(255)
        bufs[ib[3]][: B * H * W * 3] = np.clip(v, 0, 255).astype(np.uint8).reshape(-1)
    else:
        raise AssertionError(f"unknown LaMa op {k}")


def replay_lama(view, packed_weights, images_u8, masks_u8):
    """images_u8 [B,H,W,3], masks_u8 [B,H,W] -> (uint8 [B,H,W,3], buffers)."""
    bufs = {BUF_PLAN_CONST: view.consts}
    for b, n in enumerate(view.buf_elems):
        if b == LB_WEIGHTS:
            bufs[b] = np.asarray(packed_weights, dtype=np.float32)
        elif b in (LB_IN_U8, LB_MASK_U8, LB_OUT_U8):
            bufs[b] = np.zeros(n, dtype=np.uint8)
        else:
            bufs[b] = np.zeros(n, dtype=np.float32)
    bufs[LB_IN_U8][: images_u8.size] = images_u8.reshape(-1)
    bufs[LB_MASK_U8][: masks_u8.size] = masks_u8.reshape(-1)
    with torch.no_grad():
        for info, items in view.ops:
            if info.kind == _replay.OP_GEMM:
                for it in items:
                    _replay.gemm_reference(it, info.bmode, bufs, view.tables)
            elif info.kind == OP_EW:
                ew_reference(info, bufs)
            else:
                raise AssertionError(f"unexpected op kind {info.kind} in a LaMa plan")
    return bufs[LB_OUT_U8][: images_u8.size].reshape(images_u8.shape).copy(), bufs
