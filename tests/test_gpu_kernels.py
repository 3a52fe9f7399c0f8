"""Kernel-level parity on the MI355X: every HIP kernel, called through the C-ABI, against the
descriptor semantics replayed on the CPU (tests/_replay.py) or the oracle's cv2 restatement."""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import _replay
from oracle import cv2_restate as cv2r

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _make_gemm_case(rng, M, N, K, bm, bn, bmode, splitK=1, bias=True, act=1, residual=True, alpha=1.0):
    """Random gather-GEMM problem: scattered, 16-byte aligned row/chunk offsets."""
    tilesM, tilesN = -(-M // bm), -(-N // bn)
    kc = K // 32
    Mp, Np = tilesM * bm, tilesN * bn
    # A[m,k] = Abuf[rowA[m] + colA[k/32] + k%32]
    colA = (rng.permutation(kc + 3)[:kc] * 32).astype(np.int64)
    rowstrideA = (kc + 3) * 32 + 4
    rowA = (rng.permutation(Mp + 5)[:Mp] * rowstrideA).astype(np.int64)
    rowA[M:] = rowA[0]
    Abuf = rng.standard_normal((Mp + 5) * rowstrideA + 64).astype(np.float32)
    if bmode == 0:
        colB = (rng.permutation(kc + 2)[:kc] * 32).astype(np.int64)
        rowstrideB = (kc + 2) * 32 + 8
        rowB = (rng.permutation(Np + 3)[:Np] * rowstrideB).astype(np.int64)
        rowB[N:] = rowB[0]
        Bbuf = rng.standard_normal((Np + 3) * rowstrideB + 64).astype(np.float32)
    else:
        ncc = Np // 32
        colB = (rng.permutation(ncc + 2)[:ncc] * 32).astype(np.int64)
        rowstrideB = (ncc + 2) * 32 + 4
        rowB = (rng.permutation(K + 3)[:K] * rowstrideB).astype(np.int64)
        Bbuf = rng.standard_normal((K + 3) * rowstrideB + 64).astype(np.float32)
    ncc = Np // 32
    colC = (rng.permutation(ncc + 1)[:ncc] * 32).astype(np.int64)
    rowstrideC = (ncc + 1) * 32
    rowC = (rng.permutation(Mp + 2)[:Mp] * rowstrideC).astype(np.int64)
    rowC[M:] = rowC[0]
    plane = (Mp + 2) * rowstrideC
    Cbuf = np.full(plane * splitK + 64, -7.0, dtype=np.float32)
    rowR = (rng.permutation(Mp + 2)[:Mp] * rowstrideC).astype(np.int64)
    Rbuf = rng.standard_normal(plane + 64).astype(np.float32)
    biasv = rng.standard_normal(Np).astype(np.float32)
    cps = -(-kc // splitK)
    assert (splitK - 1) * cps < kc
    partial = splitK > 1
    return SimpleNamespace(
        M=M, N=N, K=K, tilesM=tilesM, tilesN=tilesN, splitK=splitK, chunksPerSplit=cps, splitStride=plane,
        alpha=alpha, act=0 if partial else act, bmode=bmode,
        Abuf=Abuf, Bbuf=Bbuf, Cbuf=Cbuf, Rbuf=Rbuf, biasv=biasv,
        rowA=rowA, colA=colA, rowB=rowB, colB=colB, rowC=rowC, colC=colC, rowR=rowR,
        use_bias=bias and not partial, use_res=residual and not partial)


def _reference(case):
    bufs = {0: case.biasv, 1: case.Abuf, 2: case.Bbuf, 3: case.Cbuf.copy(), 4: case.Rbuf}
    tables = [case.rowA, case.colA, case.rowB, case.colB, case.rowC, case.colC, case.rowR]
    it = SimpleNamespace(M=case.M, N=case.N, K=case.K, bufA=1, bufB=2, bufC=3, bufR=4 if case.use_res else -1,
                         offA=0, offB=0, offC=0, offR=0, offBias=0 if case.use_bias else -1,
                         tRowA=0, tColA=1, tRowB=2, tColB=3, tRowC=4, tColC=5, tRowR=6,
                         splitK=case.splitK, chunksPerSplit=case.chunksPerSplit, splitStride=case.splitStride,
                         alpha=case.alpha, act=case.act)
    _replay.gemm_reference(it, case.bmode, bufs, tables)
    return bufs[3]


def _run_cases(_lib, device, cases, tile_cfg, bmode, variant=None):
    keep = []
    probs = (_lib.GGProblem * len(cases))()
    outs = []
    for i, c in enumerate(cases):
        d = {k: _dev(getattr(c, k), device) for k in ("Abuf", "Bbuf", "Cbuf", "Rbuf", "biasv")}
        t = {k: _dev(getattr(c, k).astype(np.int32), device) for k in ("rowA", "colA", "rowB", "colB", "rowC", "colC", "rowR")}
        keep.append((d, t))
        p = probs[i]
        p.A, p.B, p.C = d["Abuf"].data_ptr(), d["Bbuf"].data_ptr(), d["Cbuf"].data_ptr()
        p.bias = d["biasv"].data_ptr() if c.use_bias else None
        p.R = d["Rbuf"].data_ptr() if c.use_res else None
        p.rowA, p.colA, p.rowB, p.colB = (t[k].data_ptr() for k in ("rowA", "colA", "rowB", "colB"))
        p.rowC, p.colC = t["rowC"].data_ptr(), t["colC"].data_ptr()
        p.rowR = t["rowR"].data_ptr() if c.use_res else None
        p.M, p.N, p.K, p.tilesM, p.tilesN = c.M, c.N, c.K, c.tilesM, c.tilesN
        p.splitK, p.chunksPerSplit, p.act, p.alpha, p.splitStride = c.splitK, c.chunksPerSplit, c.act, c.alpha, c.splitStride
        outs.append(d["Cbuf"])
    torch.cuda.synchronize()
    if variant is None:
        _lib.check(_lib.lib.vsr_run_gather_gemm(probs, len(cases), tile_cfg, bmode, None))
    else:
        _lib.check(_lib.lib.vsr_run_gather_gemm_variant(probs, len(cases), tile_cfg, bmode, variant, None))
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in outs]


def _written_mask(case):
    """Cells of Cbuf the descriptor writes: (m < M, n < N) of every split-K plane."""
    m, n = np.meshgrid(np.arange(case.M), np.arange(case.N), indexing="ij")
    cell = (case.rowC[m] + case.colC[n // 32] + n % 32).reshape(-1)
    mask = np.zeros(case.Cbuf.shape[0], dtype=bool)
    for s in range(case.splitK):
        mask[cell + s * case.splitStride] = True
    return mask


def _assert_close(got, ref, K, what, case=None):
    err = np.abs(got - ref).max()
    tol = 2e-5 * np.sqrt(K) * 4 + 1e-5
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol:.3e}"
    # untouched sentinel cells must stay untouched (no out-of-tile stores)
    if case is not None:      # exact footprint (a computed value may by chance equal the sentinel)
        keep = ~_written_mask(case)
        assert np.all(ref[keep] == -7.0)
        assert np.array_equal(got[keep], ref[keep]), f"{what}: store footprint differs"
    else:
        assert np.array_equal(got == -7.0, ref == -7.0), f"{what}: store footprint differs"


@pytest.mark.parametrize("M,N,K,splitK,bias,act,res", [
    (300, 200, 96, 1, True, 1, True),
    (128, 128, 32, 1, False, 0, False),
    (60, 60, 640, 4, False, 0, False),        # split-K partial planes (coarse attention scales)
    (375, 375, 384, 3, False, 0, False),
    (1, 1, 64, 1, True, 0, True),
    (513, 257, 2304, 1, True, 1, True),       # conv-like K, ragged M/N
])
def test_gather_gemm_128x128_nk(built_lib, gpu_device, M, N, K, splitK, bias, act, res):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    c = _make_gemm_case(rng, M, N, K, 128, 128, 0, splitK, bias, act, res, alpha=0.5 if splitK == 1 else 1.0)
    got = _run_cases(built_lib, gpu_device, [c], built_lib.TILE_128x128, 0)[0]
    _assert_close(got, _reference(c), K, f"NK {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(200, 192, 96), (60, 76800 // 50, 64), (129, 960, 160), (33, 64, 32)])
def test_gather_gemm_128x128_kn(built_lib, gpu_device, M, N, K):
    rng = np.random.default_rng(M + N + K)
    c = _make_gemm_case(rng, M, N, K, 128, 128, 1, 1, False, 0, False)
    got = _run_cases(built_lib, gpu_device, [c], built_lib.TILE_128x128, 1)[0]
    _assert_close(got, _reference(c), K, f"KN {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K,splitK,bmode", [(300, 256, 2304, 1, 0), (129, 65, 64, 1, 0), (60, 60, 640, 5, 0),
                                                 (200, 192, 96, 1, 1), (1440, 960, 320, 3, 1), (33, 64, 32, 1, 1)])
def test_gather_gemm_128x64(built_lib, gpu_device, M, N, K, splitK, bmode):
    rng = np.random.default_rng(M + N + K + bmode)
    full = splitK == 1 and bmode == 0
    c = _make_gemm_case(rng, M, N, K, 128, 64, bmode, splitK, full, 1 if full else 0, full)
    got = _run_cases(built_lib, gpu_device, [c], built_lib.TILE_128x64, bmode)[0]
    _assert_close(got, _reference(c), K, f"128x64 mode{bmode} {M}x{N}x{K} split{splitK}")


def test_reduce_scatter(built_lib, gpu_device):
    rng = np.random.default_rng(8)
    M, N, ns = 333, 960, 3
    part = rng.standard_normal((ns, M, N)).astype(np.float32)
    ncc = N // 32
    colC = (rng.permutation(ncc + 2)[:ncc] * 32).astype(np.int32)
    rowC = (rng.permutation(M + 3)[:M] * ((ncc + 2) * 32)).astype(np.int32)
    out = torch.full(((M + 3) * (ncc + 2) * 32,), -7.0, device=gpu_device)
    dpart, drow, dcol = _dev(part, gpu_device), _dev(rowC, gpu_device), _dev(colC, gpu_device)   # keep alive
    rc = built_lib.lib.vsr_launch_reduce_scatter(_ptr(dpart), ns, M * N, M, N, _ptr(drow), _ptr(dcol), _ptr(out), None)
    assert rc == 0
    torch.cuda.synchronize()
    ref = np.full(out.shape[0], -7.0, np.float32)
    cols = (colC.astype(np.int64)[:, None] + np.arange(32)[None, :]).reshape(-1)
    ref[rowC.astype(np.int64)[:, None] + cols[None, :]] = (part[0] + part[1]) + part[2]
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("cfg,bn,M,N,K", [("TILE_256x32", 32, 700, 3, 64), ("TILE_256x32", 32, 256, 32, 576),
                                           ("TILE_256x64", 64, 520, 64, 576), ("TILE_256x64", 64, 1000, 64, 32)])
def test_gather_gemm_narrow_tiles(built_lib, gpu_device, cfg, bn, M, N, K):
    rng = np.random.default_rng(M + 13 * N + K)
    c = _make_gemm_case(rng, M, N, K, 256, bn, 0, 1, True, 1, N > 3)
    got = _run_cases(built_lib, gpu_device, [c], getattr(built_lib, cfg), 0)[0]
    _assert_close(got, _reference(c), K, f"{cfg} {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K,act,bias", [(700, 3, 64, 1, True), (1000, 2, 2304, 0, True), (513, 1, 288, 2, False), (300, 4, 576, 3, True),
                                             (256, 2, 32, 0, True)])
def test_gather_gemm_dot_product_kernel(built_lib, gpu_device, M, N, K, act, bias):
    """VSR_VARIANT_NARROW (csrc/gather_gemm_narrow.h): problems of at most four output columns on the 256 x 32 tile's tables -- the
    descriptor semantics of the MFMA kernels (scattered rows / chunks, alpha, bias, activation, exact store footprint), and the MFMA
    kernel's result within fp32 summation-order noise."""
    VARIANT_NARROW = 8
    rng = np.random.default_rng(M + 17 * N + K)
    c = _make_gemm_case(rng, M, N, K, 256, 32, 0, 1, bias, act, False, alpha=0.75)
    got = _run_cases(built_lib, gpu_device, [c], built_lib.TILE_256x32, 0, VARIANT_NARROW)[0]
    _assert_close(got, _reference(c), K, f"narrow {M}x{N}x{K}", case=c)
    mfma = _run_cases(built_lib, gpu_device, [c], built_lib.TILE_256x32, 0, 3)[0]
    assert np.abs(got - mfma).max() <= 2e-5 * np.sqrt(K)


def test_gather_gemm_dot_product_kernel_grouped_and_refused(built_lib, gpu_device):
    """several problems of one launch (N = 2 and N = 3 together run on the 4-column instance); a residual, split-K or N = 5 is refused"""
    rng = np.random.default_rng(99)
    cases = [_make_gemm_case(rng, 333, 2, 288, 256, 32, 0, 1, True, 0, False), _make_gemm_case(rng, 700, 3, 576, 256, 32, 0, 1, True, 1, False),
             _make_gemm_case(rng, 5, 2, 64, 256, 32, 0, 1, False, 0, False)]
    for c, got in zip(cases, _run_cases(built_lib, gpu_device, cases, built_lib.TILE_256x32, 0, 8)):
        _assert_close(got, _reference(c), c.K, f"narrow grouped {c.M}x{c.N}x{c.K}", case=c)
    for bad in (_make_gemm_case(rng, 300, 5, 64, 256, 32, 0), _make_gemm_case(rng, 300, 2, 64, 256, 32, 0, 1, True, 0, True),
                _make_gemm_case(rng, 300, 2, 128, 256, 32, 0, 2)):
        with pytest.raises(built_lib.VsrError):
            _run_cases(built_lib, gpu_device, [bad], built_lib.TILE_256x32, 0, 8)


@pytest.mark.parametrize("variant", [1, 3, 4])
@pytest.mark.parametrize("cfg,bm,bn,bmode,M,N,K,splitK", [
    ("TILE_128x128", 128, 128, 0, 513, 257, 2304, 1), ("TILE_128x64", 128, 64, 0, 300, 256, 576, 1),
    ("TILE_128x128", 128, 128, 0, 375, 375, 384, 3), ("TILE_256x64", 256, 64, 0, 520, 64, 576, 1),
    ("TILE_256x32", 256, 32, 0, 700, 3, 64, 1),
    ("TILE_128x128", 128, 128, 1, 200, 192, 96, 1), ("TILE_128x64", 128, 64, 1, 1440, 960, 320, 3),
    ("TILE_128x128", 128, 128, 1, 129, 960, 4800 // 32 * 32, 1),
])
def test_gather_gemm_every_variant(built_lib, gpu_device, variant, cfg, bm, bn, bmode, M, N, K, splitK):
    """All four kernel variants implement the same descriptor semantics (v4 = split-half f16 MFMA: operands
    here are O(1), far inside the fp16 range; its error is ~2^-22 relative per product)."""
    rng = np.random.default_rng(variant * 1000 + M + N + K)
    full = splitK == 1 and bmode == 0
    c = _make_gemm_case(rng, M, N, K, bm, bn, bmode, splitK, full, 1 if full else 0, full and N > 3)
    got = _run_cases(built_lib, gpu_device, [c], getattr(built_lib, cfg), bmode, variant)[0]
    _assert_close(got, _reference(c), K, f"v{variant} {cfg} mode{bmode} {M}x{N}x{K}", case=c)


@pytest.mark.parametrize("cfg,bm,bn,M,N,K,splitK,act,res", [
    ("TILE_128x64", 128, 64, 1300, 128, 1152, 1, 1, True), ("TILE_128x128", 128, 128, 700, 512, 2304, 1, 2, True),
    ("TILE_256x32", 256, 32, 900, 32, 576, 1, 1, False), ("TILE_128x64", 128, 64, 375, 375, 384, 3, 0, False),
    ("TILE_128x64", 128, 64, 4321, 64, 64, 1, 3, True),
])
def test_one_workgroup_per_tile_kernel_equals_the_persistent_one(built_lib, gpu_device, cfg, bm, bn, M, N, K, splitK, act, res):
    """variant 1 (one workgroup per tile) and variant 3 (persistent, LDS-DMA, pipelined tiles) run the same fp32 MFMA sequence per output
    element -- chunks in order, 16 k-steps per chunk, the same epilogue: THE SAME BITS.  What lets an engine choose the kernel per problem
    (short / few-tile problems on variant 1: flow_engine.hip `thin_variant`, ocr_det_nhwc.thin_variant) without touching a parity claim."""
    rng = np.random.default_rng(M + N + K)
    full = splitK == 1
    c = _make_gemm_case(rng, M, N, K, bm, bn, 0, splitK, full, act if full else 0, res and full)
    one = _run_cases(built_lib, gpu_device, [c], getattr(built_lib, cfg), 0, 1)[0]
    per = _run_cases(built_lib, gpu_device, [c], getattr(built_lib, cfg), 0, 3)[0]
    assert np.array_equal(one, per)
    _assert_close(one, _reference(c), K, f"v1 {cfg} {M}x{N}x{K}", case=c)


def _to_split_inplace(buf, starts):
    """fp32 -> split format on the 32-float chunks starting at ``starts`` (see include/vsr_hip.h, precision 2)."""
    starts = np.unique(np.asarray(starts, dtype=np.int64))
    idx = starts[:, None] + np.arange(32)[None, :]
    v = buf[idx]
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    buf[idx] = np.ascontiguousarray(np.concatenate([hi, lo], axis=1)).view(np.float32)


def _starts(row, col):
    return (np.asarray(row, np.int64)[:, None] + np.asarray(col, np.int64)[None, :]).reshape(-1)


@pytest.mark.parametrize("out_split", [0, 1])
@pytest.mark.parametrize("cfg,bm,bn,bmode,M,N,K,splitK", [
    ("TILE_128x128", 128, 128, 0, 513, 257, 2304, 1), ("TILE_128x64", 128, 64, 0, 300, 256, 576, 1),
    ("TILE_128x128", 128, 128, 0, 375, 375, 384, 3), ("TILE_256x64", 256, 64, 0, 520, 64, 576, 1),
    ("TILE_256x32", 256, 32, 0, 700, 3, 64, 1),
    ("TILE_128x128", 128, 128, 1, 200, 192, 96, 1), ("TILE_128x64", 128, 64, 1, 1440, 960, 320, 3),
    ("TILE_128x64", 128, 64, 1, 129, 960, 4800 // 32 * 32, 1),
    ("TILE_256x256", 256, 256, 0, 513, 257, 2304, 1), ("TILE_256x256", 256, 256, 0, 700, 320, 96, 1),     # gather_gemm_f16_v7<1>
])
def test_gather_gemm_split_format(built_lib, gpu_device, out_split, cfg, bm, bn, bmode, M, N, K, splitK):
    """Variant 5: A, B and R arrive in split format ([32 fp16 hi | 32 fp16 lo] per 32-float chunk); C leaves in
    split format when act carries VSR_ACT_OUT_SPLIT, as plain fp32 otherwise (split-K planes always fp32)."""
    if out_split and splitK > 1:
        pytest.skip("partial planes are plain fp32")
    rng = np.random.default_rng(5000 + M + N + K + out_split)
    full = splitK == 1
    c = _make_gemm_case(rng, M, N, K, bm, bn, bmode, splitK, full, 1 if full else 0, full and N > 3)
    ref = _reference(c)
    sentinel = c.Cbuf.copy()
    _to_split_inplace(c.Abuf, _starts(c.rowA, c.colA))
    _to_split_inplace(c.Bbuf, _starts(c.rowB, c.colB))
    if c.use_res:
        _to_split_inplace(c.Rbuf, _starts(c.rowR[:M], c.colC))
    if out_split:
        c.act |= 0x100                                   # VSR_ACT_OUT_SPLIT
    got = _run_cases(built_lib, gpu_device, [c], getattr(built_lib, cfg), bmode, 5)[0]
    what = f"v5 {cfg} mode{bmode} {M}x{N}x{K} out_split={out_split}"
    if not out_split:
        _assert_close(got, ref, K, what, case=c)
        return
    m, n = np.meshgrid(np.arange(M), np.arange(N), indexing="ij")
    chunk = c.rowC[m] + c.colC[n // 32]
    hi_i, lo_i = 2 * chunk + n % 32, 2 * chunk + 32 + n % 32
    g16 = got.view(np.uint16)
    val = g16[hi_i].view(np.float16).astype(np.float32) + g16[lo_i].view(np.float16).astype(np.float32)
    err = np.abs(val - ref[chunk + n % 32]).max()
    tol = 2e-5 * np.sqrt(K) * 4 + 1e-5
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol:.3e}"
    s16 = sentinel.view(np.uint16)
    rest = g16.copy()
    rest[hi_i] = s16[hi_i]
    rest[lo_i] = s16[lo_i]
    assert np.array_equal(rest, s16), f"{what}: store footprint differs"


@pytest.mark.parametrize("out_split", [0, 1])
@pytest.mark.parametrize("cfg,bm,bn,M,N,K,splitK", [
    ("TILE_128x128", 128, 128, 513, 257, 2304, 1), ("TILE_128x64", 128, 64, 300, 256, 576, 1),
    ("TILE_128x64", 128, 64, 260, 130, 96, 1),               # 3 chunks: one full pair + a lone chunk
    ("TILE_128x64", 128, 64, 200, 64, 352, 1),               # 11 chunks
    ("TILE_128x128", 128, 128, 375, 375, 384, 3),            # split-K: 4 chunks per slice
    ("TILE_128x128", 128, 128, 140, 140, 480, 2),            # split-K with odd slices: 8 + 7 chunks
    ("TILE_256x64", 256, 64, 520, 64, 576, 1),
    ("TILE_128x64", 128, 64, 4800, 256, 32 * 150, 1),        # 150 chunks: crosses a 128-chunk super-block
])
def test_gather_gemm_fp16_operands(built_lib, gpu_device, out_split, cfg, bm, bn, M, N, K, splitK):
    """Variant 6 on NK problems = gather_gemm_f16_v6: the fp16 hi halves of split-format operands alone, two K chunks per LDS
    stage.  With operands that are exactly representable in fp16 the result is the fp32-accumulated product itself."""
    if out_split and splitK > 1:
        pytest.skip("partial planes are plain fp32")
    rng = np.random.default_rng(6000 + M + N + K + out_split)
    full = splitK == 1
    c = _make_gemm_case(rng, M, N, K, bm, bn, 0, splitK, full, 1 if full else 0, full)
    c.Abuf[:] = c.Abuf.astype(np.float16).astype(np.float32)
    c.Bbuf[:] = c.Bbuf.astype(np.float16).astype(np.float32)
    ref = _reference(c)
    sentinel = c.Cbuf.copy()
    _to_split_inplace(c.Abuf, _starts(c.rowA, c.colA))
    _to_split_inplace(c.Bbuf, _starts(c.rowB, c.colB))
    if c.use_res:
        _to_split_inplace(c.Rbuf, _starts(c.rowR[:M], c.colC))
    if out_split:
        c.act |= 0x100
    got = _run_cases(built_lib, gpu_device, [c], getattr(built_lib, cfg), 0, 6)[0]
    what = f"v6 {cfg} {M}x{N}x{K} splitK={splitK} out_split={out_split}"
    if not out_split:
        _assert_close(got, ref, K, what, case=c)
        return
    m, n = np.meshgrid(np.arange(M), np.arange(N), indexing="ij")
    chunk = c.rowC[m] + c.colC[n // 32]
    hi_i, lo_i = 2 * chunk + n % 32, 2 * chunk + 32 + n % 32
    g16 = got.view(np.uint16)
    val = g16[hi_i].view(np.float16).astype(np.float32) + g16[lo_i].view(np.float16).astype(np.float32)
    err = np.abs(val - ref[chunk + n % 32]).max()
    tol = 2e-5 * np.sqrt(K) * 4 + 1e-5
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol:.3e}"
    s16 = sentinel.view(np.uint16)
    rest = g16.copy()
    rest[hi_i] = s16[hi_i]
    rest[lo_i] = s16[lo_i]
    assert np.array_equal(rest, s16), f"{what}: store footprint differs"


def _split_case(rng, M, N, K, bm, bn, representable=True):
    """an NK problem on split-format operands (bias, LeakyReLU, residual) and its fp32 reference"""
    c = _make_gemm_case(rng, M, N, K, bm, bn, 0, 1, True, 1, True)
    if representable:
        c.Abuf[:] = c.Abuf.astype(np.float16).astype(np.float32)
        c.Bbuf[:] = c.Bbuf.astype(np.float16).astype(np.float32)
    ref = _reference(c) if representable else None
    _to_split_inplace(c.Abuf, _starts(c.rowA, c.colA))
    _to_split_inplace(c.Bbuf, _starts(c.rowB, c.colB))
    _to_split_inplace(c.Rbuf, _starts(c.rowR[:M], c.colC))
    return c, ref


@pytest.mark.parametrize("out_split", [0, 1])
@pytest.mark.parametrize("M,N,K,tilesM", [
    (300, 256, 576, None),
    (513, 257, 2304, None),            # N not a multiple of 32: the element-wise epilogue; a partial last M tile
    (260, 130, 96, None),              # 3 chunks: one full pair + a lone chunk; N = 130
    (1000, 512, 320, 8),               # tile height 128 (tilesM given: 8 tiles of roundup32(125) rows), two N tiles
    (200, 64, 352, 7),                 # tile height 32: the second wave row owns no block
    (448, 256, 128, 2),                # tile height 224: 4 + 3 blocks over the two wave rows
    (700, 320, 64, 3),                 # second N tile 64 columns wide
    (4800, 256, 32 * 150, None),       # 150 chunks: crosses a 128-chunk super-block
])
def test_gather_gemm_fp16_256x256(built_lib, gpu_device, out_split, M, N, K, tilesM):
    """TILE_256x256 (variant 6, NK) = gather_gemm_f16_v7: 8 waves, 256 x 256 tile of dynamic height, output turned through LDS.  With
    fp16-representable operands the result is the fp32-accumulated product itself; the store footprint is exact."""
    rng = np.random.default_rng(7000 + M + N + K + out_split)
    c, ref = _split_case(rng, M, N, K, 256, 256)
    if tilesM is not None:
        c.tilesM = tilesM
    sentinel = c.Cbuf.copy()
    if out_split:
        c.act |= 0x100
    got = _run_cases(built_lib, gpu_device, [c], built_lib.TILE_256x256, 0, 6)[0]
    what = f"v7 256x256 {M}x{N}x{K} tilesM={tilesM} out_split={out_split}"
    if not out_split:
        _assert_close(got, ref, K, what, case=c)
        return
    m, n = np.meshgrid(np.arange(M), np.arange(N), indexing="ij")
    chunk = c.rowC[m] + c.colC[n // 32]
    hi_i, lo_i = 2 * chunk + n % 32, 2 * chunk + 32 + n % 32
    g16 = got.view(np.uint16)
    val = g16[hi_i].view(np.float16).astype(np.float32) + g16[lo_i].view(np.float16).astype(np.float32)
    err = np.abs(val - ref[chunk + n % 32]).max()
    tol = 2e-5 * np.sqrt(K) * 4 + 1e-5
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol:.3e}"
    s16 = sentinel.view(np.uint16)
    rest = g16.copy()
    rest[hi_i] = s16[hi_i]
    rest[lo_i] = s16[lo_i]
    assert np.array_equal(rest, s16), f"{what}: store footprint differs"


@pytest.mark.parametrize("out_split", [0, 1])
@pytest.mark.parametrize("variant", [6, 5])
def test_gather_gemm_fp16_256x256_equals_128x64(built_lib, gpu_device, out_split, variant):
    """Same k order, same MFMAs in the same order: on ANY split-format operands (not only fp16-representable ones) the 256 x 256 kernel
    writes the bytes the 128 x 64 kernel of the same variant (6: fp16 operands, 5: split-half, three MFMAs per product) writes --
    also as two problems of one launch."""
    rng = np.random.default_rng(7100 + out_split)
    outs = {}
    for cfg, bm, bn in (("TILE_128x64", 128, 64), ("TILE_256x256", 256, 256)):
        rng = np.random.default_rng(7100 + out_split)
        cases = []
        for (M, N, K) in ((1300, 256, 576), (900, 384, 960)):
            c, _ = _split_case(rng, M, N, K, 256, 256, representable=False)
            c.tilesM, c.tilesN = -(-M // bm), -(-N // bn)
            if out_split:
                c.act |= 0x100
            cases.append(c)
        outs[cfg] = _run_cases(built_lib, gpu_device, cases, getattr(built_lib, cfg), 0, variant)
    for a, b in zip(outs["TILE_128x64"], outs["TILE_256x256"]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("variant", [6, 5])
@pytest.mark.parametrize("M,N,K,splitK,tilesM", [(1440, 960, 1472, 3, None), (700, 320, 480, 2, 5)])
def test_gather_gemm_256x256_split_k(built_lib, gpu_device, variant, M, N, K, splitK, tilesM):
    """split-K on the 256 x 256 kernel: every (tile, slice) writes its own fp32 partial plane, as on the 128 x 64 kernel"""
    outs = {}
    for cfg, bm, bn in (("TILE_128x64", 128, 64), ("TILE_256x256", 256, 256)):
        rng = np.random.default_rng(7300 + M + splitK)
        c = _make_gemm_case(rng, M, N, K, 256, 256, 0, splitK, False, 0, False)
        _to_split_inplace(c.Abuf, _starts(c.rowA, c.colA))
        _to_split_inplace(c.Bbuf, _starts(c.rowB, c.colB))
        c.tilesM, c.tilesN = -(-M // bm), -(-N // bn)
        if cfg == "TILE_256x256" and tilesM is not None:
            c.tilesM = tilesM
        outs[cfg] = _run_cases(built_lib, gpu_device, [c], getattr(built_lib, cfg), 0, variant)[0]
    assert np.array_equal(outs["TILE_128x64"].view(np.uint32), outs["TILE_256x256"].view(np.uint32))


def test_kn_to_nk_split(built_lib, gpu_device):
    """vsr_launch_kn_to_nk_split: a gathered KN operand in split format becomes the dense NK operand dst[n * ld + k] in split format"""
    rng = np.random.default_rng(7400)
    K, N, ld = 96, 64, 128
    rowstride = (N // 32 + 2) * 32
    rowB = (rng.permutation(K + 3)[:K] * rowstride).astype(np.int64)
    colB = (rng.permutation(N // 32 + 2)[: N // 32] * 32).astype(np.int64)
    src = rng.integers(0, 2 ** 32, size=(K + 3) * rowstride + 64, dtype=np.uint32)
    dst = np.zeros(N * ld + 32, dtype=np.uint32)
    d_src, d_dst = _dev(src.view(np.float32), gpu_device), _dev(dst.view(np.float32), gpu_device)
    t_row, t_col = _dev(rowB.astype(np.int32), gpu_device), _dev(colB.astype(np.int32), gpu_device)
    built_lib.check(built_lib.lib.vsr_launch_kn_to_nk_split(_ptr(d_src), _ptr(t_row), _ptr(t_col), K, N, C.c_int64(ld), _ptr(d_dst), None))
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy().view(np.uint16)
    s16 = src.view(np.uint16)
    want = np.zeros_like(got)
    k, n = np.meshgrid(np.arange(K), np.arange(N), indexing="ij")
    src_chunk = rowB[k] + colB[n // 32]                       # float offset of the source chunk
    dst_chunk = n * ld + 32 * (k // 32)
    want[2 * dst_chunk + k % 32] = s16[2 * src_chunk + n % 32]                 # hi halves
    want[2 * dst_chunk + 32 + k % 32] = s16[2 * src_chunk + 32 + n % 32]       # lo halves
    assert np.array_equal(got, want)


def test_to_split_bit_exact(built_lib, gpu_device):
    rng = np.random.default_rng(8)
    x = (rng.standard_normal(32 * 1000) * np.exp(rng.uniform(-8, 8, 32 * 1000))).astype(np.float32)
    dx = _dev(x, gpu_device)
    out = torch.empty_like(dx)
    assert built_lib.lib.vsr_launch_to_split(_ptr(dx), _ptr(out), x.size, None) == 0
    torch.cuda.synchronize()
    ref = x.copy()
    _to_split_inplace(ref, np.arange(0, x.size, 32))
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_gather_gemm_grouped_launch(built_lib, gpu_device):
    """Several problems of different shape in one grid (as the 4 attention scales are)."""
    rng = np.random.default_rng(99)
    cases = [_make_gemm_case(rng, 60, 60, 1280, 128, 128, 0, 8, False, 0, False),
             _make_gemm_case(rng, 375, 375, 384, 128, 128, 0, 2, False, 0, False),
             _make_gemm_case(rng, 700, 700, 96, 128, 128, 0, 1, False, 0, False),
             _make_gemm_case(rng, 130, 130, 32, 128, 128, 0, 1, True, 1, True)]
    gots = _run_cases(built_lib, gpu_device, cases, built_lib.TILE_128x128, 0)
    for i, (g, c) in enumerate(zip(gots, cases)):
        _assert_close(g, _reference(c), c.K, f"group member {i}")


def test_gather_gemm_asymmetric_identity(built_lib, gpu_device):
    """A = I with an asymmetric B catches transposed / mis-mapped MFMA fragments exactly."""
    M = N = K = 128
    rng = np.random.default_rng(1)
    c = _make_gemm_case(rng, M, N, K, 128, 128, 0, 1, False, 0, False)
    eye = np.eye(M, dtype=np.float32)
    colsA = (c.colA[:, None] + np.arange(32)[None, :]).reshape(-1)
    c.Abuf[:] = 0
    c.Abuf[c.rowA[:M, None] + colsA[None, :]] = eye
    colsB = (c.colB[:, None] + np.arange(32)[None, :]).reshape(-1)
    Bm = (np.arange(K)[:, None] * 1000 + np.arange(N)[None, :]).astype(np.float32)      # B[k,n]
    c.Bbuf[c.rowB[:N, None] + colsB[None, :]] = Bm.T
    got = _run_cases(built_lib, gpu_device, [c], built_lib.TILE_128x128, 0)[0]
    colsC = (c.colC[:, None] + np.arange(32)[None, :]).reshape(-1)[:N]
    assert np.array_equal(got[c.rowC[:M, None] + colsC[None, :]], Bm)


def test_softmax_rows(built_lib, gpu_device):
    rng = np.random.default_rng(5)
    specs = [(60, 60, 5), (375, 375, 2), (1441, 1441, 1), (7, 4800, 1)]
    probs = (built_lib.SMProblem * len(specs))()
    keep, refs = [], []
    for i, (M, N, ns) in enumerate(specs):
        ld = -(-N // 32) * 32
        S = (rng.standard_normal((ns, M, ld)) * 3).astype(np.float32)
        scale = 1.0 / np.sqrt(960.0)
        dS, dP = _dev(S, gpu_device), torch.full((M, ld), 9.0, device=gpu_device)
        keep.append((dS, dP))
        p = probs[i]
        p.S, p.P, p.M, p.N, p.ldS, p.ldP, p.nsplit, p.scale, p.splitStride = dS.data_ptr(), dP.data_ptr(), M, N, ld, ld, ns, scale, M * ld
        acc = S[0].copy()
        for s in range(1, ns):
            acc = acc + S[s]
        ref = np.zeros((M, ld), np.float32)
        ref[:, :N] = torch.softmax(torch.from_numpy(acc[:, :N] * np.float32(scale)), -1).numpy()
        refs.append(ref)
    built_lib.check(built_lib.lib.vsr_run_softmax(probs, len(specs), None))
    torch.cuda.synchronize()
    for (dS, dP), ref in zip(keep, refs):
        got = dP.cpu().numpy()
        assert np.abs(got - ref).max() < 2e-6
        assert np.abs(got.sum(1) - 1).max() < 1e-5


def _tables(ssize, dsize, clamp, device):
    ofs, ic, fc = cv2r.linear_tables(ssize, dsize, clamp)
    return (_dev(ofs.astype(np.int32), device), _dev(ic.reshape(-1), device), _dev(fc.reshape(-1), device)), (ofs, ic, fc)


def test_host_cv2_tables_match_oracle(built_lib):
    for ssize, dsize, clamp in [(1280, 640, 1), (1920, 640, 1), (3840, 640, 1), (240, 120, 0), (360, 120, 0),
                                (640, 1920, 1), (120, 360, 0), (640, 1280, 1), (1000, 640, 1), (187, 120, 0), (640, 853, 1)]:
        ofs = np.zeros(dsize, np.int32)
        ic = np.zeros(2 * dsize, np.int16)
        fc = np.zeros(2 * dsize, np.float32)
        built_lib.check(built_lib.lib.vsr_cv2_linear_tables(ssize, dsize, clamp, ofs.ctypes.data_as(C.c_void_p),
                                                           ic.ctypes.data_as(C.c_void_p), fc.ctypes.data_as(C.c_void_p)))
        o2, i2, f2 = cv2r.linear_tables(ssize, dsize, bool(clamp))
        assert np.array_equal(ofs, o2) and np.array_equal(ic, i2.reshape(-1)) and np.array_equal(fc, f2.reshape(-1))


@pytest.mark.parametrize("sw,sh", [(1280, 240), (1920, 360), (3840, 720), (1000, 187), (852, 159)])
def test_resize_u8_down_bit_exact(built_lib, gpu_device, sw, sh):
    rng = np.random.default_rng(sw)
    n, H = 3, sh + 20
    frames = rng.integers(0, 256, size=(n, H, sw, 3), dtype=np.uint8)
    ymin = 13
    (xo, xa, _), _ = _tables(sw, 640, True, gpu_device)
    (yo, ya, _), _ = _tables(sh, 120, False, gpu_device)
    src = _dev(frames, gpu_device)
    dst = torch.zeros((n, 120, 640, 3), dtype=torch.uint8, device=gpu_device)
    idx = _dev(np.array([2, 0, 1], np.int32), gpu_device)
    rc = built_lib.lib.vsr_launch_resize_u8(C.c_void_p(src.data_ptr() + ymin * sw * 3), H * sw * 3, sw * 3, sw, sh, _ptr(dst),
                                            640, 120, n, 3, _ptr(idx), _ptr(xo), _ptr(xa), _ptr(yo), _ptr(ya), None)
    assert rc == 0
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    for o, f in enumerate([2, 0, 1]):
        ref = cv2r.resize_linear(frames[f, ymin:ymin + sh], (640, 120))
        assert np.array_equal(got[o], ref)


def test_resize_u8_single_channel_mask(built_lib, gpu_device):
    """sttn-det resizes the 0/255 mask strip with the frames (frame stride 0 replicates it per frame)."""
    sw, sh, n = 1920, 533, 3
    mask = np.zeros((sh + 10, sw), np.uint8)
    mask[100:300, 291:1632] = 255
    (xo, xa, _), _ = _tables(sw, 432, True, gpu_device)
    (yo, ya, _), _ = _tables(sh, 240, False, gpu_device)
    src = _dev(mask, gpu_device)
    dst = torch.zeros((n, 240, 432), dtype=torch.uint8, device=gpu_device)
    rc = built_lib.lib.vsr_launch_resize_u8(C.c_void_p(src.data_ptr() + 5 * sw), 0, sw, sw, sh, _ptr(dst), 432, 240, n, 1, None,
                                            _ptr(xo), _ptr(xa), _ptr(yo), _ptr(ya), None)
    assert rc == 0
    torch.cuda.synchronize()
    ref = cv2r.resize_linear(mask[5:5 + sh, :, None], (432, 240))[:, :, 0]
    got = dst.cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], ref)
    assert 0 < ((ref > 0) & (ref < 255)).sum(), "the resized mask has interpolated edge values"


def test_norm_im2col_exact(built_lib, gpu_device):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(2, 120, 640, 3), dtype=np.uint8)
    out = torch.zeros((2 * 60 * 320, 32), device=gpu_device)
    dimg = _dev(img, gpu_device)
    assert built_lib.lib.vsr_launch_norm_im2col(_ptr(dimg), 120, 640, 2, _ptr(out), 0, None, None) == 0
    torch.cuda.synchronize()
    bufs = {0: img.reshape(-1), 1: np.zeros(2 * 60 * 320 * 32, np.float32)}
    _replay.norm_im2col_reference(SimpleNamespace(H=120, W=640, n=2, buf_src=0, buf_dst=1), bufs)
    assert np.array_equal(out.cpu().numpy().reshape(-1), bufs[1])


@pytest.mark.parametrize("H,W,Cc,hs,hd", [(30, 160, 256, 2, 1), (60, 320, 64, 0, 1)])
def test_upsample2x(built_lib, gpu_device, H, W, Cc, hs, hd):
    rng = np.random.default_rng(H)
    n = 2
    src = np.zeros((n, H + 2 * hs, W + 2 * hs, Cc), np.float32)
    src[:, hs:hs + H, hs:hs + W] = rng.standard_normal((n, H, W, Cc)).astype(np.float32)
    dst = torch.zeros((n, 2 * H + 2 * hd, 2 * W + 2 * hd, Cc), device=gpu_device)
    dsrc = _dev(src, gpu_device)
    assert built_lib.lib.vsr_launch_upsample2x(_ptr(dsrc), H, W, Cc, hs, _ptr(dst), hd, n, None) == 0
    torch.cuda.synchronize()
    x = torch.from_numpy(src[:, hs:hs + H, hs:hs + W]).permute(0, 3, 1, 2)
    ref = torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1).numpy()
    got = dst.cpu().numpy()
    assert np.abs(got[:, hd:hd + 2 * H, hd:hd + 2 * W] - ref).max() < 2e-6
    if hd:
        assert not got[:, :hd].any() and not got[:, :, :hd].any() and not got[:, -hd:].any() and not got[:, :, -hd:].any()


def test_decode_out_and_average(built_lib, gpu_device):
    rng = np.random.default_rng(4)
    n, pix, L = 3, 640 * 8, 5
    y = (rng.standard_normal((n, pix, 32)) * 1.2).astype(np.float32)
    comp0 = rng.integers(0, 256, size=(L, pix, 3)).astype(np.float32)
    comp = _dev(comp0, gpu_device)
    fidx, first = np.array([4, 1, 2], np.int32), np.array([1, 0, 0], np.int32)
    dy, dfidx, dfirst = _dev(y, gpu_device), _dev(fidx, gpu_device), _dev(first, gpu_device)      # keep alive
    assert built_lib.lib.vsr_launch_decode_out(_ptr(dy), 32, pix, n, _ptr(dfidx), _ptr(dfirst), _ptr(comp), None, None, None) == 0
    torch.cuda.synchronize()
    bufs = {0: y.reshape(-1), 1: comp0.reshape(-1).copy()}
    _replay.decode_out_reference(SimpleNamespace(n=n, pix=pix, ldy=32, buf_src=0, buf_dst=1, t_frame_idx=0, t_first=1),
                                 bufs, [fidx.astype(np.int64), first.astype(np.int64)])
    got, ref = comp.cpu().numpy().reshape(-1), bufs[1]
    d = np.abs(got - ref)
    assert d.max() <= 1.0 and (d > 0).mean() < 1e-3      # tanhf ulp differences at truncation boundaries only
    assert np.array_equal(got.reshape(L, -1)[[0, 3]], comp0.reshape(L, -1)[[0, 3]])


@pytest.mark.parametrize("W,sh", [(1280, 240), (1920, 360), (853, 159)])
def test_upscale_blend_bit_exact(built_lib, gpu_device, W, sh):
    from oracle.sttn_auto import STTNInpaintOracle

    rng = np.random.default_rng(W)
    n, H, ymin = 3, sh + 30, 17
    frames = rng.integers(0, 256, size=(n, H, W, 3), dtype=np.uint8)
    mask = np.zeros((H, W), np.uint8)
    mask[ymin + 20: ymin + sh - 10, W // 6: W - W // 5] = 1
    comp_u8 = rng.integers(0, 256, size=(n, 120, 640, 3)).astype(np.float32)
    comp_f = comp_u8 * np.float32(0.5) + rng.integers(0, 256, size=(n, 120, 640, 3)).astype(np.float32) * np.float32(0.5)
    isf = np.array([0, 1, 1], np.int32)
    comp = np.where(isf[:, None, None, None] == 1, comp_f, comp_u8).astype(np.float32)
    (xo, xa, xf), _ = _tables(640, W, True, gpu_device)
    (yo, ya, yf), _ = _tables(120, sh, False, gpu_device)
    dfr = _dev(frames, gpu_device)
    dmask = _dev(mask, gpu_device)
    dcomp, disf = _dev(comp, gpu_device), _dev(isf, gpu_device)                                 # keep alive
    rc = built_lib.lib.vsr_launch_upscale_blend(
        _ptr(dcomp), 640, 120, _ptr(disf), C.c_void_p(dfr.data_ptr() + ymin * W * 3), H * W * 3,
        W * 3, None, C.c_void_p(dmask.data_ptr() + ymin * W), W, W, sh, n, _ptr(xo), _ptr(xa), _ptr(xf), _ptr(yo), _ptr(ya), _ptr(yf), None)
    assert rc == 0
    torch.cuda.synchronize()
    got = dfr.cpu().numpy()
    o = STTNInpaintOracle.__new__(STTNInpaintOracle)
    for i in range(n):
        ref = frames[i].copy()
        c = comp[i] if isf[i] else comp[i].astype(np.uint8)
        o.blend_strip(ref, c, mask[:, :, None], (ymin, ymin + sh, 0, W), W, sh)
        assert np.array_equal(got[i], ref), f"frame {i} (float path={bool(isf[i])})"


# ---- fused attention: row maxima out of the score GEMM, exp + row sums inside the P.V GEMM -------------------------------
def _ordered_to_float(u):
    """inverse of the kernels' monotone unsigned image of a float (csrc/gather_gemm.hip f32_ordered)"""
    u = u.astype(np.uint32)
    pos = (u & np.uint32(0x80000000)) != 0
    bits = np.where(pos, u & np.uint32(0x7FFFFFFF), ~u)
    return bits.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("M,N,K,tile", [(320, 320, 960, "128x64"), (130, 75, 64, "128x64"), (300, 200, 96, "128x128"), (4800, 4800, 64, "128x64")])
def test_score_gemm_leaves_row_maxima(built_lib, gpu_device, M, N, K, tile):
    """VSR_ACT_ROW_MAX on the persistent NK kernel: C as without the flag, and max_n C[m][n] of every row in the side array
    (atomic max over the N tiles of an ordered-uint image; rows beyond M untouched)"""
    rng = np.random.default_rng(M + N + K)
    bm, bn = (128, 64) if tile == "128x64" else (128, 128)
    cfg = built_lib.TILE_128x64 if tile == "128x64" else built_lib.TILE_128x128
    c = _make_gemm_case(rng, M, N, K, bm, bn, 0, 1, False, 0, False, alpha=0.125)
    ref = _reference(c)
    mp = c.tilesM * bm
    rowmax = torch.zeros(mp + 8, dtype=torch.int32, device=gpu_device)
    c.act = 0x400
    keep = []
    probs = (built_lib.GGProblem * 1)()
    d = {k: _dev(getattr(c, k), gpu_device) for k in ("Abuf", "Bbuf", "Cbuf")}
    t = {k: _dev(getattr(c, k).astype(np.int32), gpu_device) for k in ("rowA", "colA", "rowB", "colB", "rowC", "colC")}
    p = probs[0]
    p.A, p.B, p.C, p.bias, p.R = d["Abuf"].data_ptr(), d["Bbuf"].data_ptr(), d["Cbuf"].data_ptr(), None, rowmax.data_ptr()
    p.rowA, p.colA, p.rowB, p.colB, p.rowC, p.colC = (t[k].data_ptr() for k in ("rowA", "colA", "rowB", "colB", "rowC", "colC"))
    p.rowR = None
    p.M, p.N, p.K, p.tilesM, p.tilesN = c.M, c.N, c.K, c.tilesM, c.tilesN
    p.splitK, p.chunksPerSplit, p.act, p.alpha, p.splitStride = 1, c.chunksPerSplit, c.act, c.alpha, c.splitStride
    built_lib.check(built_lib.lib.vsr_run_gather_gemm_variant(probs, 1, cfg, 0, 3, None))
    torch.cuda.synchronize()
    got = d["Cbuf"].cpu().numpy()
    _assert_close(got, ref, K, f"scores {M}x{N}x{K}", c)
    m, n = np.meshgrid(np.arange(M), np.arange(N), indexing="ij")
    Cmat = got[c.rowC[m] + c.colC[n // 32] + n % 32]
    rm = rowmax.cpu().numpy()
    assert np.array_equal(_ordered_to_float(rm[:M]), Cmat.max(axis=1)), "row maxima are not the maxima of the stored scores"
    assert not rm[M:].any(), "rows beyond M were touched"


@pytest.mark.parametrize("M,K,N,splitK,tile", [(320, 320, 960, 1, "128x64"), (4800, 4800, 64, 3, "128x64"), (200, 96, 130, 1, "128x64"),
                                               (640, 640, 320, 2, "128x128")])
def test_pv_gemm_exponentiates_its_scores(built_lib, gpu_device, M, K, N, splitK, tile):
    """VSR_ACT_A_EXP on the KN kernel (variant 1 | VSR_VARIANT_A_EXP): C = (2^(A - rowmax) . B) / row sums without a probability matrix --
    normalised in the epilogue (splitK 1) or left as partial planes + partial row sums (splitK > 1); a problem without the flag
    in the same launch runs as the plain KN kernel"""
    rng = np.random.default_rng(M + N + K + splitK)
    bm, bn = (128, 64) if tile == "128x64" else (128, 128)
    cfg = built_lib.TILE_128x64 if tile == "128x64" else built_lib.TILE_128x128
    c = _make_gemm_case(rng, M, N, K, bm, bn, 1, splitK, False, 0, False)
    c.Abuf *= 3.0                                                 # scores with a spread: 2^x spans ~2^25
    plain = _make_gemm_case(rng, 150, 70, 64, bm, bn, 1, 1, False, 0, False)
    mp = c.tilesM * bm
    m_idx, k_idx = np.meshgrid(np.arange(M), np.arange(K), indexing="ij")
    S = c.Abuf[c.rowA[m_idx] + c.colA[k_idx // 32] + k_idx % 32]
    rowmax_f = np.zeros(mp, np.float32)
    rowmax_f[:M] = S.max(axis=1)
    b = rowmax_f.view(np.uint32)
    enc = np.where((b & np.uint32(0x80000000)) != 0, ~b, b | np.uint32(0x80000000)).astype(np.uint32)
    enc[M:] = 0
    rowmax = torch.from_numpy(enc.view(np.int32)).to(gpu_device)
    lsum = torch.full((splitK * mp + 8,), -3.0, dtype=torch.float32, device=gpu_device)
    probs = (built_lib.GGProblem * 2)()
    keep, outs = [], []
    for i, cc in enumerate((c, plain)):
        d = {k: _dev(getattr(cc, k), gpu_device) for k in ("Abuf", "Bbuf", "Cbuf")}
        t = {k: _dev(getattr(cc, k).astype(np.int32), gpu_device) for k in ("rowA", "colA", "rowB", "colB", "rowC", "colC")}
        keep.append((d, t))
        p = probs[i]
        p.A, p.B, p.C = d["Abuf"].data_ptr(), d["Bbuf"].data_ptr(), d["Cbuf"].data_ptr()
        p.bias, p.R = (rowmax.data_ptr(), lsum.data_ptr() if splitK > 1 else None) if i == 0 else (None, None)
        p.rowA, p.colA, p.rowB, p.colB, p.rowC, p.colC = (t[k].data_ptr() for k in ("rowA", "colA", "rowB", "colB", "rowC", "colC"))
        p.rowR = None
        p.M, p.N, p.K, p.tilesM, p.tilesN = cc.M, cc.N, cc.K, cc.tilesM, cc.tilesN
        p.splitK, p.chunksPerSplit, p.alpha, p.splitStride = cc.splitK, cc.chunksPerSplit, 1.0, cc.splitStride
        p.act = 0x800 if i == 0 else 0
        outs.append(d["Cbuf"])
    built_lib.check(built_lib.lib.vsr_run_gather_gemm_variant(probs, 2, cfg, 1, 1 | 0x100, None))
    torch.cuda.synchronize()
    _assert_close(outs[1].cpu().numpy(), _reference(plain), plain.K, "plain problem beside a fused one", plain)
    got = outs[0].cpu().numpy()
    k2, n2 = np.meshgrid(np.arange(K), np.arange(N), indexing="ij")
    Bm = torch.from_numpy(c.Bbuf[c.rowB[k2] + c.colB[n2 // 32] + n2 % 32]).double()
    E = torch.exp2(torch.from_numpy(S).double() - torch.from_numpy(rowmax_f[:M]).double()[:, None])
    want = (E @ Bm) / E.sum(dim=1, keepdim=True)
    m2, n3 = np.meshgrid(np.arange(M), np.arange(N), indexing="ij")
    cells = c.rowC[m2] + c.colC[n3 // 32] + n3 % 32
    if splitK == 1:
        out = torch.from_numpy(got[cells]).double()
    else:
        ls = lsum.cpu().numpy()
        parts = sum(torch.from_numpy(got[cells + s * c.splitStride]).double() for s in range(splitK))
        ltot = sum(torch.from_numpy(ls[s * mp: s * mp + M]).double() for s in range(splitK))
        assert np.allclose(ltot.numpy(), E.sum(dim=1).numpy(), rtol=2e-6), "partial row sums"
        assert (ls[splitK * mp:] == -3.0).all()
        out = parts / ltot[:, None]
    err = (out - want).abs().max().item()
    assert err <= 2e-6 * max(1.0, want.abs().max().item()) * np.sqrt(K) / 4 + 2e-6, err
    keepmask = ~_written_mask(c)
    assert np.array_equal(got[keepmask], c.Cbuf[keepmask]), "store footprint differs"
