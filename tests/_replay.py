"""CPU replay of an engine plan (test infrastructure).

Reads the op list, symbolic buffers and offset tables the engine would run for
STTNInpaint.inpaint(L) through the vsr_plan_* C entry points and executes them with numpy /
torch-CPU.  This checks every table, descriptor and schedule decision of the host engine
against the oracle without a GPU; the GPU tests then only have to establish that each kernel
implements the same descriptor semantics.
"""
import ctypes as C

import numpy as np
import torch

OP_NORM_IM2COL, OP_GEMM, OP_SOFTMAX, OP_UPSAMPLE2X, OP_DECODE_OUT, OP_REDUCE_SCATTER = range(6)
BUF_WEIGHTS, BUF_IN_U8 = 0, 1


class PlanView:
    def __init__(self, _lib, engine, L, plan_ptr=None, rows=None, cols=None):
        """STTN: PlanView(_lib, engine, L[, rows=(lo, hi): the decoder on these model rows only]).  Any other plan: pass the
        vsr_plan_t pointer (L = length of the counts array, 0 = none)."""
        self._lib = _lib
        lib = _lib.lib
        self.p = plan_ptr if plan_ptr is not None else C.c_void_p()
        if plan_ptr is None and rows is not None and cols is not None:
            _lib.check(lib.vsr_plan_create_box(engine.handle, L, int(rows[0]), int(rows[1]), int(cols[0]), int(cols[1]), C.byref(self.p)))
        elif plan_ptr is None and rows is not None:
            _lib.check(lib.vsr_plan_create_rows(engine.handle, L, int(rows[0]), int(rows[1]), C.byref(self.p)))
        elif plan_ptr is None:
            _lib.check(lib.vsr_plan_create(engine.handle, L, C.byref(self.p)))
        self.L = L
        self.buf_elems = [lib.vsr_plan_buffer_elems(self.p, b) for b in range(lib.vsr_plan_num_buffers(self.p))]
        self.tables = []
        for t in range(lib.vsr_plan_num_tables(self.p)):
            n = lib.vsr_plan_table_len(self.p, t)
            arr = np.empty(n, dtype=np.int32)
            _lib.check(lib.vsr_plan_table_copy(self.p, t, arr.ctypes.data_as(C.c_void_p)))
            self.tables.append(arr.astype(np.int64))
        self.ops = []
        for i in range(lib.vsr_plan_num_ops(self.p)):
            info = _lib.VsrOpInfo()
            _lib.check(lib.vsr_plan_op(self.p, i, C.byref(info)))
            items = []
            for j in range(info.nitems):
                if info.kind == OP_GEMM:
                    it = _lib.VsrGemmInfo()
                    _lib.check(lib.vsr_plan_op_gemm(self.p, i, j, C.byref(it)))
                else:
                    it = _lib.VsrSoftmaxInfo()
                    _lib.check(lib.vsr_plan_op_softmax(self.p, i, j, C.byref(it)))
                items.append(it)
            self.ops.append((info, items))
        counts = np.zeros(L, dtype=np.int32)
        if L:
            _lib.check(lib.vsr_plan_counts(self.p, counts.ctypes.data_as(C.c_void_p)))
        self.counts = counts
        self.flops = lib.vsr_plan_flops(self.p)

    def close(self):
        if self.p:
            self._lib.lib.vsr_plan_destroy(self.p)
            self.p = None


def _cols(table, n):
    """32-element chunk offsets -> per-element offsets of the first n elements."""
    full = (table[:, None] + np.arange(32, dtype=np.int64)[None, :]).reshape(-1)
    return full[:n]


def _gather(buf, base, rows, coltab, n):
    """buf[base + rows[:, None] + cols(coltab, n)[None, :]] -- by whole 32-element chunks (a chunk is 32 contiguous elements: one index per
    chunk through a sliding-window view instead of one per element; the same values, ~20x faster than the element-wise fancy index)"""
    nfull = n // 32
    rows = np.asarray(rows, dtype=np.int64)
    parts = []
    if nfull:
        win = np.lib.stride_tricks.sliding_window_view(buf, 32)
        parts.append(win[base + rows[:, None] + np.asarray(coltab[:nfull], dtype=np.int64)[None, :]].reshape(len(rows), nfull * 32))
    if n % 32:
        tail = np.asarray(coltab[nfull], dtype=np.int64) + np.arange(n % 32, dtype=np.int64)
        parts.append(buf[base + rows[:, None] + tail[None, :]])
    return parts[0] if len(parts) == 1 else np.concatenate(parts, axis=1)


def _scatter(buf, base, rows, coltab, n, values):
    """buf[base + rows[:, None] + cols(coltab, n)[None, :]] = values, by whole chunks (see _gather)"""
    nfull = n // 32
    rows = np.asarray(rows, dtype=np.int64)
    values = np.ascontiguousarray(values, dtype=buf.dtype)
    if nfull:
        win = np.lib.stride_tricks.sliding_window_view(buf, 32, writeable=True)
        win[base + rows[:, None] + np.asarray(coltab[:nfull], dtype=np.int64)[None, :]] = values[:, :nfull * 32].reshape(len(rows), nfull, 32)
    if n % 32:
        tail = np.asarray(coltab[nfull], dtype=np.int64) + np.arange(n % 32, dtype=np.int64)
        buf[base + rows[:, None] + tail[None, :]] = values[:, nfull * 32:]


def gemm_reference(it, bmode, bufs, tables, tile_m=128):
    """Execute one gather-GEMM descriptor exactly as include/vsr_hip.h defines it."""
    M, N, K = it.M, it.N, it.K
    A = bufs[it.bufA]
    B = bufs[it.bufB]
    rowA = tables[it.tRowA][:M]
    Am = torch.from_numpy(_gather(A, it.offA, rowA, tables[it.tColA], K))
    if bmode == 0:      # NK
        rowB = tables[it.tRowB][:N]
        Bm = torch.from_numpy(_gather(B, it.offB, rowB, tables[it.tColB], K)).t()      # K x N
    else:               # KN
        rowB = tables[it.tRowB][:K]
        Bm = torch.from_numpy(_gather(B, it.offB, rowB, tables[it.tColB], N))          # K x N
    rowC = tables[it.tRowC][:M]
    tColC = tables[it.tColC]
    Cbuf = bufs[it.bufC]
    if it.act & 0x800:                     # VSR_ACT_A_EXP: A holds scores; the product is softmax(A) . B, normalised here or by the reduce op
        Am = torch.exp2(Am - Am.max(dim=1, keepdim=True).values)    # the scores carry log2(e) / sqrt(D)
        if it.splitK > 1:
            ldl = it.tilesM * tile_m
            for s in range(it.splitK):
                k0 = s * it.chunksPerSplit * 32
                k1 = min(K, k0 + it.chunksPerSplit * 32)
                _scatter(Cbuf, it.offC + s * it.splitStride, rowC, tColC, N, (Am[:, k0:k1] @ Bm[k0:k1, :]).numpy())
                bufs[it.bufR][it.offR + s * ldl: it.offR + s * ldl + M] = Am[:, k0:k1].sum(dim=1).numpy()
        else:
            _scatter(Cbuf, it.offC, rowC, tColC, N, ((Am @ Bm) / Am.sum(dim=1, keepdim=True)).numpy())
        return
    if it.splitK > 1:
        for s in range(it.splitK):
            k0 = s * it.chunksPerSplit * 32
            k1 = min(K, k0 + it.chunksPerSplit * 32)
            part = (Am[:, k0:k1] @ Bm[k0:k1, :]) * it.alpha
            _scatter(Cbuf, it.offC + s * it.splitStride, rowC, tColC, N, part.numpy())
        return
    acc = (Am @ Bm) * it.alpha
    if it.offBias >= 0:
        acc = acc + torch.from_numpy(bufs[BUF_WEIGHTS][it.offBias:it.offBias + N])[None, :]
    if it.act & 0xff == 1:
        acc = torch.nn.functional.leaky_relu(acc, 0.2)
    elif it.act & 0xff == 2:
        acc = torch.relu(acc)
    elif it.act & 0xff == 3:
        acc = torch.nn.functional.leaky_relu(acc, 0.1)
    if it.act & 0x400:                     # VSR_ACT_ROW_MAX: bufR receives the row maxima (ordered-uint images; the replay's A_EXP takes its own)
        pass
    elif it.bufR >= 0:
        rowR = tables[it.tRowR][:M]
        acc = acc + torch.from_numpy(_gather(bufs[it.bufR], it.offR, rowR, tColC, N))
        if it.act & 0x200:                 # VSR_ACT_POST_RELU
            acc = torch.relu(acc)
    _scatter(Cbuf, it.offC, rowC, tColC, N, acc.numpy())


def softmax_reference(it, bufs):
    S = bufs[it.bufS]
    acc = np.zeros((it.M, it.N), dtype=np.float32)
    for s in range(it.nsplit):
        plane = S[it.offS + s * it.splitStride: it.offS + s * it.splitStride + it.M * it.ldS].reshape(it.M, it.ldS)
        acc = acc + plane[:, :it.N] if s else plane[:, :it.N].copy()
    p = torch.softmax(torch.from_numpy(acc * np.float32(it.scale)), dim=-1).numpy()
    out = np.zeros((it.M, it.ldP), dtype=np.float32)
    out[:, :it.N] = p
    bufs[it.bufP][it.offP: it.offP + it.M * it.ldP] = out.reshape(-1)


def upsample_reference(info, bufs):
    H, W, Cc, hs, hd, n = info.H, info.W, info.C, info.halo_src, info.halo_dst, info.n
    Hs, Ws = H + 2 * hs, W + 2 * hs
    Hd, Wd = 2 * H + 2 * hd, 2 * W + 2 * hd
    src = bufs[info.buf_src][: n * Hs * Ws * Cc].reshape(n, Hs, Ws, Cc)[:, hs:hs + H, hs:hs + W, :]
    x = torch.from_numpy(np.ascontiguousarray(src)).permute(0, 3, 1, 2)
    y = torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    dst = bufs[info.buf_dst][: n * Hd * Wd * Cc].reshape(n, Hd, Wd, Cc)
    lo, hi = int(info.ipar[1]), int(info.ipar[2])       # output rows written (a plan with a decoder row range); 0, 0 = all
    if hi <= lo:
        lo, hi = 0, 2 * H
    dst[:, hd + lo:hd + hi, hd:hd + 2 * W, :] = y.permute(0, 2, 3, 1).numpy()[:, lo:hi]


def norm_im2col_reference(info, bufs):
    ih, iw, n = info.H, info.W, info.n
    img = bufs[info.buf_src][: n * ih * iw * 3].reshape(n, ih, iw, 3)
    x = torch.from_numpy(np.ascontiguousarray(img[..., ::-1])).permute(0, 3, 1, 2).float().div(255) * 2 - 1
    if getattr(info, "premask", 0):
        m = bufs[info.buf_mask][: n * ih * iw].reshape(n, 1, ih, iw)
        x = x * torch.from_numpy((m < 128).astype(np.float32))
    cols = torch.nn.functional.unfold(x, kernel_size=3, padding=1, stride=2)       # n, 27 (c,ky,kx), oh*ow
    oh, ow = ih // 2, iw // 2
    cols = cols.view(n, 3, 9, oh * ow).permute(0, 3, 2, 1).reshape(n * oh * ow, 27)   # k = tap*3 + c
    out = np.zeros((n * oh * ow, 32), dtype=np.float32)
    out[:, :27] = cols.numpy()
    bufs[info.buf_dst][: out.size] = out.reshape(-1)


def decode_out_reference(info, bufs, tables):
    n, pix, ldy = info.n, info.pix, info.ldy
    if getattr(info, "W", 0) > 0:           # rows are 2x4 pixel blocks of a W-wide image, columns (dy, dx, channel): the blocked output conv
        mw, mh = info.W, pix // info.W
        blk = bufs[info.buf_src][: n * (pix // 8) * ldy].reshape(n, mh // 2, mw // 4, ldy)[..., :24].reshape(n, mh // 2, mw // 4, 2, 4, 3)
        y = blk.transpose(0, 1, 3, 2, 4, 5).reshape(n, pix, 3)
    else:
        y = bufs[info.buf_src][: n * pix * ldy].reshape(n, pix, ldy)[:, :, :3]
    v = torch.tanh(torch.from_numpy(np.ascontiguousarray(y)))
    v = ((v + 1) / 2).numpy() * 255
    img = v.astype(np.uint8).astype(np.float32)
    comp = bufs[info.buf_dst]
    det_mask = bufs[info.buf_mask] if getattr(info, "buf_mask", -1) >= 0 else None
    idx = tables[info.t_frame_idx]
    first = tables[info.t_first]
    p0, p1 = 0, pix                                      # pixels decoded: whole image rows [ipar[1], ipar[2]) of a W-wide image
    if getattr(info, "W", 0) > 0 and int(info.ipar[2]) > int(info.ipar[1]):
        p0, p1 = int(info.ipar[1]) * info.W, int(info.ipar[2]) * info.W
    for i in range(n):
        if det_mask is not None:
            f = int(idx[i])
            b = (det_mask[f * pix:(f + 1) * pix] > 0)[:, None]
            rgb = bufs[BUF_IN_U8][f * pix * 3:(f + 1) * pix * 3].reshape(pix, 3)[:, ::-1].astype(np.float32)
            img[i] = np.where(b, img[i], rgb)
        base = int(idx[i]) * pix * 3
        sl = slice(base + p0 * 3, base + p1 * 3)
        new = img[i].reshape(-1)[p0 * 3:p1 * 3]
        if first[i]:
            comp[sl] = new
        else:
            comp[sl] = comp[sl] * np.float32(0.5) + new * np.float32(0.5)


def reduce_scatter_reference(info, bufs, tables):
    M, N = info.M, info.N
    part = bufs[info.buf_src]
    acc = None
    for s in range(info.nsplit):
        o = info.off_src + s * info.split_stride
        plane = part[o: o + M * N].reshape(M, N)
        acc = plane.copy() if acc is None else acc + plane
    if info.ibuf[0] >= 0:                  # planes of a VSR_ACT_A_EXP product: divide by the total of the splits' row sums
        lbuf, ldl = bufs[info.ibuf[0]], info.ipar[0]
        l = None
        for s in range(info.nsplit):
            part_l = lbuf[info.ioff[0] + s * ldl: info.ioff[0] + s * ldl + M]
            l = part_l.copy() if l is None else l + part_l
        acc = acc / l[:, None]
    rowC = tables[info.t_rowC][:M]
    colC = _cols(tables[info.t_colC], N)
    bufs[info.buf_dst][info.off_dst + rowC[:, None] + colC[None, :]] = acc


BUF_MASK_U8 = 22


def replay(view, packed_weights, frames_u8, masks_u8=None):
    """frames_u8: [L,mh,mw,3] uint8 BGR (+ masks_u8 [L,mh,mw] for sttn-det) -> (comp f32 [L,mh,mw,3] RGB, counts)."""
    bufs = []
    for b, n in enumerate(view.buf_elems):
        if b == BUF_WEIGHTS:
            bufs.append(np.asarray(packed_weights, dtype=np.float32))
        elif b == BUF_IN_U8:
            a = np.zeros(n, dtype=np.uint8)
            a[: frames_u8.size] = frames_u8.reshape(-1)
            bufs.append(a)
        elif b == BUF_MASK_U8:
            a = np.zeros(n, dtype=np.uint8)
            if masks_u8 is not None:
                a[: masks_u8.size] = masks_u8.reshape(-1)
            bufs.append(a)
        else:
            bufs.append(np.zeros(n, dtype=np.float32))
    with torch.no_grad():
        for info, items in view.ops:
            if info.kind == OP_GEMM:
                for it in items:
                    gemm_reference(it, info.bmode, bufs, view.tables, tile_m={0: 128, 1: 256, 2: 256, 3: 128, 4: 256}[info.tile_cfg])
            elif info.kind == OP_SOFTMAX:
                for it in items:
                    softmax_reference(it, bufs)
            elif info.kind == OP_UPSAMPLE2X:
                upsample_reference(info, bufs)
            elif info.kind == OP_NORM_IM2COL:
                norm_im2col_reference(info, bufs)
            elif info.kind == OP_DECODE_OUT:
                decode_out_reference(info, bufs, view.tables)
            elif info.kind == OP_REDUCE_SCATTER:
                reduce_scatter_reference(info, bufs, view.tables)
            else:
                raise AssertionError(f"unknown op kind {info.kind}")
    comp = bufs[20][: frames_u8.size].reshape(frames_u8.shape).copy()     # BUF_COMP
    return comp, view.counts, bufs
