"""torch-tensor convenience layer over the C handle of libvsr_hip.so.

PyTorch is plumbing here: device memory (tensors own the HBM buffers handed to the C-ABI as
raw pointers), the current HIP stream and, for multi-GPU, torch.distributed.  All arithmetic
of the hot path happens inside the library's hand-written kernels.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib, switches
from ._lib import check, lib


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    if lib.vsr_device_count() <= 0 or not torch.cuda.is_available():
        raise _lib.VsrError(_lib.VSR_ERR_NOGPU, "no HIP device visible: the MI355X path has no CPU fallback")


# vsr_sttn_set_precision modes: exact fp32 MFMA | split-half f16 MFMA (guarded) | the same on split-format
# tensors | fp16 operands + fp32 accumulation (BASELINE.json's "fp16 MFMA path")
PRECISION_MODES = {"f32": 0, "split": 1, "split-format": 2, "f16": 3}
# vsr_{raft,rfc,pp,lama}_set_precision: exact fp32 | split-half (fp16 hi/lo pairs, 22 bits) | fp16 operands, fp32 accumulation
FLOW_PRECISION_MODES = {"f32": 0, "split": 1, "f16": 2}


class AccuracyGuard:
    """Self-check of the reduced-precision arithmetics (round 6).

    The range guard inside the kernels sees values that leave the fp16 range; it cannot see ACCURACY.  tests/test_gpu_weight_sweep.py
    showed what that misses: on weights with sharp attention rows (synth.py "peaked") or heavy tails the fp16-operand mode stays in range
    and lands at 47 / 28 dB (STTN) or 17 dB (ProPainter generator) against the oracle where the benign draw gives 60 -- every GEMM in
    front of the attention feeds 11-bit errors into logits that are 16x larger.  Weights are a property of the checkpoint, not of the
    frame, so the check is cheap: the FIRST unit of work an engine sees in a guarded mode (and every VSR_F16_SELFCHECK_EVERY-th after it,
    default 256; 0 = off) is also run in exact fp32, and when the two differ by more than VSR_F16_SELFCHECK_DB (default 50 dB, the bar
    of BASELINE.json) the caller gets the exact result and the engine is DEMOTED one step along its chain (f16 -> split-format ->
    f32; the new mode is checked on its own first unit).  `demotions` counts them and fallbacks() includes them."""

    def _guard_init(self, precision, chain):
        self.precision = precision
        self._guard_chain = chain
        self._guard_every = int(os.environ.get("VSR_F16_SELFCHECK_EVERY", "256"))
        self._guard_db = float(os.environ.get("VSR_F16_SELFCHECK_DB", "50"))
        self._guard_calls = 0
        self.demotions = 0
        self.guard_log = []                     # (mode, psnr dB) of every check

    def _guard_due(self):
        due = self.precision != "f32" and self._guard_every > 0 and self._guard_calls % self._guard_every == 0
        self._guard_calls += 1
        return due

    def _guard_exact(self):
        """switch to exact fp32 for the reference run of a check; returns the guarded mode for _guard_verdict"""
        mode = self.precision
        self._apply_precision("f32")
        return mode

    def _guard_verdict(self, mode, got, want, where, peak):
        """True: `got` (computed in `mode`) is within the bar of `want` (exact) over `where` (bool tensor or None = everywhere); the
        engine is back in `mode`.  False: the engine has been demoted; the caller hands out `want`."""
        d = (got.float() - want.float())
        if where is not None:
            d = d[where]
        mse = float((d * d).mean().item()) if d.numel() else 0.0
        psnr = float("inf") if mse == 0.0 else 20.0 * float(np.log10(peak / np.sqrt(mse)))
        if not np.isfinite(mse):
            psnr = 0.0
        self.guard_log.append((mode, round(psnr, 2) if np.isfinite(psnr) else psnr))
        if psnr >= self._guard_db:
            self._apply_precision(mode)
            return True
        self.demotions += 1
        self._apply_precision(self._guard_chain[mode])
        self._guard_calls = 0                   # the mode it fell to is checked on its own first unit
        return False

    def set_precision(self, precision):
        """the caller's choice of arithmetic; its first unit of work is checked against exact fp32 (see above)"""
        self._apply_precision(precision)
        self._guard_calls = 0


class SttnEngine(AccuracyGuard):
    """One STTN generator resident on one GPU (weights + workspace), bound to the caller's stream."""

    def __init__(self, state_dict, variant="auto", device=0, neighbor_stride=None, ref_length=None, precision=None):
        self.variant = variant
        self._h = C.c_void_p()
        check(lib.vsr_sttn_create(_lib.VARIANT[variant], C.byref(self._h)))
        try:
            for key, val in state_dict.items():
                arr = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
                arr = np.ascontiguousarray(arr, dtype=np.float32)
                shape = (C.c_int64 * arr.ndim)(*arr.shape)
                check(lib.vsr_sttn_set_param(self._h, key.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim))
            if device is not None and device >= 0:
                require_gpu()
            self.device_index = -1 if device is None else int(device)
            check(lib.vsr_sttn_finalize(self._h, self.device_index))
            self._guard_init(precision or {"1": "split", "s": "split", "2": "split-format", "3": "f16"}.get(os.environ.get("VSR_PRECISION", "0")[:1], "f32"),
                             {"f16": "split-format", "split-format": "f32", "split": "f32"})
            if precision is not None:       # "f32" (exact fp32 MFMA) | "split" (split-half f16 MFMA, guarded)
                check(lib.vsr_sttn_set_precision(self._h, PRECISION_MODES[precision]))
            if neighbor_stride is not None or ref_length is not None:
                mw, mh, ns, rl = self.geometry()
                check(lib.vsr_sttn_set_window(self._h, neighbor_stride or ns, ref_length or rl))
        except Exception:
            lib.vsr_sttn_destroy(self._h)
            self._h = None
            raise
        self.device = torch.device("cuda", self.device_index) if self.device_index >= 0 else torch.device("cpu")

    def close(self):
        if getattr(self, "_h", None):
            lib.vsr_sttn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def _apply_precision(self, precision):
        check(lib.vsr_sttn_set_precision(self._h, PRECISION_MODES[precision]))
        self.precision = precision

    def set_lanes(self, lanes):
        """1: every op of a chunk on the caller's stream; n (default 2, up to 4): sliding window w on stream w % n (same results)"""
        check(lib.vsr_sttn_set_lanes(self._h, int(lanes)))

    def fallbacks(self):
        """units redone in exact fp32: the kernels' range guard + the accuracy guard's demotions (AccuracyGuard)"""
        return int(lib.vsr_sttn_fallbacks(self._h)) + self.demotions

    def geometry(self):
        a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.vsr_sttn_geometry(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def flops(self, L, reference=False):
        """FLOPs of one inpaint(L) call as contracted here; reference=True: as the reference's modules compute it (the rows of the
        last block that nothing reads included)"""
        v = (lib.vsr_sttn_flops_reference if reference else lib.vsr_sttn_flops)(self._h, int(L))
        if v < 0:
            raise _lib.VsrError(_lib.VSR_ERR_ARG, _lib.last_error())
        return v

    def packed_weights(self):
        n = lib.vsr_sttn_packed_weights(self._h, None, 0)
        out = np.empty(n, dtype=np.float32)
        lib.vsr_sttn_packed_weights(self._h, out.ctypes.data_as(C.c_void_p), n)
        return out

    # ---- hot path -------------------------------------------------------------------------
    def inpaint(self, frames_dev):
        """STTNInpaint.inpaint: frames_dev uint8 [L,mh,mw,3] BGR on the GPU -> (comp f32 [L,mh,mw,3] RGB, counts)."""
        if not self._guard_due():
            return self._inpaint(frames_dev)
        got, counts = self._inpaint(frames_dev)
        mode = self._guard_exact()
        want, _ = self._inpaint(frames_dev)
        return (got if self._guard_verdict(mode, got, want, None, 255.0) else want), counts

    def _inpaint(self, frames_dev):
        assert frames_dev.dtype == torch.uint8 and frames_dev.is_cuda and frames_dev.is_contiguous()
        L = frames_dev.shape[0]
        comp = torch.empty(frames_dev.shape, dtype=torch.float32, device=frames_dev.device)
        counts = np.zeros(L, dtype=np.int32)
        with torch.cuda.device(frames_dev.device):
            check(lib.vsr_sttn_inpaint(self._h, C.c_void_p(frames_dev.data_ptr()), L, C.c_void_p(comp.data_ptr()),
                                       counts.ctypes.data_as(C.c_void_p), _stream_ptr()))
        return comp, counts

    def mask_rows(self, mask_dev, areas):
        """int32 [n_areas, 2]: for every area the rows [lo, hi) of its strip that hold the mask's set pixels (lo = hi = 0: none) --
        the promise vsr_sttn_auto_chunk_rows takes.  Read off the device mask once per mask (one row-flag reduction and a
        download of H bytes) and kept while the same tensor object stays unmodified (its version counter)."""
        ar = np.asarray(areas, dtype=np.int32).reshape(-1, 4)

        def from_flags(flags):
            rows = np.zeros((ar.shape[0], 2), dtype=np.int32)
            for k, (ymin, ymax, _, _) in enumerate(ar):
                nz = np.flatnonzero(flags[int(ymin):int(ymax)])
                if nz.size:
                    rows[k] = (int(nz[0]), int(nz[-1]) + 1)
            return rows

        if isinstance(mask_dev, np.ndarray):             # the caller's host copy of the same mask: no device round trip at all
            m = mask_dev.reshape(mask_dev.shape[0], mask_dev.shape[1], -1)
            return from_flags((m != 0).any(axis=(1, 2)))
        ent = getattr(self, "_mask_rows_cache", None)
        if ent is not None and ent[0]() is mask_dev and ent[1] == mask_dev._version and np.array_equal(ent[2], ar):
            return ent[3]
        import weakref

        H, W = int(mask_dev.shape[0]), int(mask_dev.shape[1])
        rows = from_flags(mask_dev.reshape(H, W).ne(0).any(dim=1).cpu().numpy())
        self._mask_rows_cache = (weakref.ref(mask_dev), mask_dev._version, ar.copy(), rows)
        return rows

    def mask_cols(self, mask, areas):
        """int32 [n_areas, 2]: the frame columns [lo, hi) that hold the set pixels of every area's strip (0, 0: none) -- the second
        half of the promise vsr_sttn_auto_chunk_box takes (honoured with VSR_DECODE_COLS=1, the default since round 5)."""
        ar = np.asarray(areas, dtype=np.int32).reshape(-1, 4)
        if not isinstance(mask, np.ndarray):             # device mask: one download per (mask object, version, areas), like mask_rows
            ent = getattr(self, "_mask_cols_cache", None)
            if ent is not None and ent[0]() is mask and ent[1] == mask._version and np.array_equal(ent[2], ar):
                return ent[3]
        cols = np.zeros((ar.shape[0], 2), dtype=np.int32)
        for k, (ymin, ymax, _, _) in enumerate(ar):
            if isinstance(mask, np.ndarray):
                flags = (mask[int(ymin):int(ymax)].reshape(int(ymax - ymin), mask.shape[1], -1) != 0).any(axis=(0, 2))
            else:
                flags = mask[int(ymin):int(ymax)].reshape(int(ymax - ymin), int(mask.shape[1]), -1).ne(0).any(dim=2).any(dim=0).cpu().numpy()
            nz = np.flatnonzero(flags)
            if nz.size:
                cols[k] = (int(nz[0]), int(nz[-1]) + 1)
        if not isinstance(mask, np.ndarray):
            import weakref

            self._mask_cols_cache = (weakref.ref(mask), mask._version, ar.copy(), cols)
        return cols

    @staticmethod
    def _check_mask_host(mask_host, mask_dev):
        """the promise about the mask's rows / columns is read off the caller's host copy: it must be THE mask (a stale or differently
        cropped copy would make the decoder skip rows the blend still reads).  Shape always; content with VSR_DEBUG_MASK_HOST=1."""
        if mask_host is None:
            return
        if tuple(mask_host.shape[:2]) != tuple(mask_dev.shape[:2]):
            raise ValueError(f"mask_host {tuple(mask_host.shape[:2])} is not the device mask's shape {tuple(mask_dev.shape[:2])}")
        if os.environ.get("VSR_DEBUG_MASK_HOST") == "1":
            h = np.asarray(mask_host).reshape(mask_host.shape[0], mask_host.shape[1], -1)[:, :, 0] != 0
            d = mask_dev.reshape(int(mask_dev.shape[0]), int(mask_dev.shape[1]), -1)[:, :, 0].ne(0).cpu().numpy()
            if not np.array_equal(h, d):
                raise ValueError("mask_host differs from the device mask")

    def chunk_flops(self, L, mask_dev, areas):
        """FLOPs of one auto_chunk call on this mask: every area's plan decodes only the rows its mask rows are resized from"""
        ar = np.asarray(areas, dtype=np.int32).reshape(-1, 4)
        total = 0.0
        with_cols = switches.on("VSR_DECODE_COLS")
        cols = self.mask_cols(mask_dev, ar) if with_cols else np.zeros((ar.shape[0], 2), dtype=np.int32)
        for (ymin, ymax, _, _), (lo, hi), (c0, c1) in zip(ar, self.mask_rows(mask_dev, ar), cols):
            if hi > lo and os.environ.get("VSR_DECODE_ROWS", "1") != "0":
                a, b = C.c_int32(), C.c_int32()
                check(lib.vsr_sttn_decode_rows(self._h, int(ymax - ymin), int(lo), int(hi), C.byref(a), C.byref(b)))
                if c1 > c0:
                    ca, cb = C.c_int32(), C.c_int32()
                    check(lib.vsr_sttn_decode_cols(self._h, int(mask_dev.shape[1]), int(c0), int(c1), C.byref(ca), C.byref(cb)))
                    v = lib.vsr_sttn_flops_box(self._h, int(L), a.value, b.value, ca.value, cb.value)
                else:
                    v = lib.vsr_sttn_flops_rows(self._h, int(L), a.value, b.value)
            else:
                v = lib.vsr_sttn_flops(self._h, int(L))
            if v < 0:
                raise _lib.VsrError(_lib.VSR_ERR_ARG, _lib.last_error())
            total += v
        return total

    def _guarded_in_place(self, raw, frames_dev):
        """an in-place unit (chunk / batch) under the accuracy guard: raw(t) works on the uint8 frame tensor t"""
        if not self._guard_due():
            return raw(frames_dev)
        src = frames_dev.clone()
        raw(frames_dev)
        mode = self._guard_exact()
        exact = src.clone()
        raw(exact)
        if not self._guard_verdict(mode, frames_dev, exact, exact != src, 255.0):
            frames_dev.copy_(exact)
        return frames_dev

    def auto_chunk(self, frames_dev, mask_dev, areas, sel=None, decode_rows=True, mask_host=None):
        """One chunk of STTNAutoInpaint.__call__, in place on frames_dev uint8 [L,H,W,3] BGR.  The rows of every strip that hold the
        mask go along (mask_rows): the decoder then computes only what the blend reads -- same frames (vsr_sttn_auto_chunk_rows)."""
        return self._guarded_in_place(lambda t: self._auto_chunk(t, mask_dev, areas, sel, decode_rows, mask_host), frames_dev)

    def _auto_chunk(self, frames_dev, mask_dev, areas, sel=None, decode_rows=True, mask_host=None):
        assert frames_dev.dtype == torch.uint8 and frames_dev.is_cuda and frames_dev.is_contiguous()
        assert mask_dev.dtype == torch.uint8 and mask_dev.is_cuda and mask_dev.is_contiguous()
        L, H, W, _ = frames_dev.shape
        assert tuple(mask_dev.shape[:2]) == (H, W)
        ar = np.ascontiguousarray(np.asarray(areas, dtype=np.int32).reshape(-1, 4))
        sel_arr = None if sel is None else np.ascontiguousarray(np.asarray(sel, dtype=np.int32))
        # decode_rows=False: no promise about the mask, the whole model-resolution image is decoded (tests compare the two)
        # (mask_host: the caller's numpy copy of the mask, when it has one -- the rows are then read off it)
        self._check_mask_host(mask_host, mask_dev)
        rows = np.ascontiguousarray(self.mask_rows(mask_dev if mask_host is None else mask_host, ar)) if decode_rows else np.zeros((ar.shape[0], 2), dtype=np.int32)
        if decode_rows and switches.on("VSR_DECODE_COLS"):
            cols = np.ascontiguousarray(self.mask_cols(mask_dev if mask_host is None else mask_host, ar))
            with torch.cuda.device(frames_dev.device):
                check(lib.vsr_sttn_auto_chunk_box(
                    self._h, C.c_void_p(frames_dev.data_ptr()), L, H, W, C.c_void_p(mask_dev.data_ptr()), ar.shape[0],
                    ar.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p), cols.ctypes.data_as(C.c_void_p),
                    None if sel_arr is None else sel_arr.ctypes.data_as(C.c_void_p),
                    0 if sel_arr is None else int(sel_arr.size), _stream_ptr()))
            return frames_dev
        with torch.cuda.device(frames_dev.device):
            check(lib.vsr_sttn_auto_chunk_rows(
                self._h, C.c_void_p(frames_dev.data_ptr()), L, H, W, C.c_void_p(mask_dev.data_ptr()), ar.shape[0],
                ar.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p),
                None if sel_arr is None else sel_arr.ctypes.data_as(C.c_void_p),
                0 if sel_arr is None else int(sel_arr.size), _stream_ptr()))
        return frames_dev

    def det_inpaint(self, frames_dev, masks_dev):
        """STTNDetInpaint.inpaint: frames uint8 [L,240,432,3] BGR + resized masks uint8 [L,240,432] -> (comp, counts)."""
        if not self._guard_due():
            return self._det_inpaint(frames_dev, masks_dev)
        got, counts = self._det_inpaint(frames_dev, masks_dev)
        mode = self._guard_exact()
        want, _ = self._det_inpaint(frames_dev, masks_dev)
        return (got if self._guard_verdict(mode, got, want, None, 255.0) else want), counts

    def _det_inpaint(self, frames_dev, masks_dev):
        assert frames_dev.dtype == torch.uint8 and frames_dev.is_cuda and frames_dev.is_contiguous()
        assert masks_dev.dtype == torch.uint8 and masks_dev.is_cuda and masks_dev.is_contiguous()
        assert tuple(masks_dev.shape) == tuple(frames_dev.shape[:3])
        L = frames_dev.shape[0]
        comp = torch.empty(frames_dev.shape, dtype=torch.float32, device=frames_dev.device)
        counts = np.zeros(L, dtype=np.int32)
        with torch.cuda.device(frames_dev.device):
            check(lib.vsr_sttn_det_inpaint(self._h, C.c_void_p(frames_dev.data_ptr()), C.c_void_p(masks_dev.data_ptr()), L,
                                           C.c_void_p(comp.data_ptr()), counts.ctypes.data_as(C.c_void_p), _stream_ptr()))
        return comp, counts

    def det_batch(self, frames_dev, mask_dev, areas, decode_rows=True, mask_host=None):
        """STTNDetInpaint.__call__ on one batch, in place on frames_dev uint8 [L,H,W,3] BGR; mask_dev raw 0/255 [H,W].  The rows of
        every strip that hold the mask go along (mask_rows): the decoder computes only the model rows the prediction is taken from
        (vsr_sttn_det_batch_rows) -- same frames; decode_rows=False: no promise."""
        return self._guarded_in_place(lambda t: self._det_batch(t, mask_dev, areas, decode_rows, mask_host), frames_dev)

    def _det_batch(self, frames_dev, mask_dev, areas, decode_rows=True, mask_host=None):
        assert frames_dev.dtype == torch.uint8 and frames_dev.is_cuda and frames_dev.is_contiguous()
        assert mask_dev.dtype == torch.uint8 and mask_dev.is_cuda and mask_dev.is_contiguous()
        L, H, W, _ = frames_dev.shape
        ar = np.ascontiguousarray(np.asarray(areas, dtype=np.int32).reshape(-1, 4))
        self._check_mask_host(mask_host, mask_dev)
        rows = np.ascontiguousarray(self.mask_rows(mask_dev if mask_host is None else mask_host, ar)) if decode_rows else np.zeros((ar.shape[0], 2), dtype=np.int32)
        if decode_rows and switches.on("VSR_DECODE_COLS"):
            cols = np.ascontiguousarray(self.mask_cols(mask_dev if mask_host is None else mask_host, ar))
            with torch.cuda.device(frames_dev.device):
                check(lib.vsr_sttn_det_batch_box(self._h, C.c_void_p(frames_dev.data_ptr()), L, H, W, C.c_void_p(mask_dev.data_ptr()),
                                                 ar.shape[0], ar.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p),
                                                 cols.ctypes.data_as(C.c_void_p), _stream_ptr()))
            return frames_dev
        with torch.cuda.device(frames_dev.device):
            check(lib.vsr_sttn_det_batch_rows(self._h, C.c_void_p(frames_dev.data_ptr()), L, H, W, C.c_void_p(mask_dev.data_ptr()),
                                              ar.shape[0], ar.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p), _stream_ptr()))
        return frames_dev

    # ---- measurement ----------------------------------------------------------------------
    def timing(self, enable=True):
        """True / 1: HIP events around every op; 2: only around the launches of the dominant gather-GEMM symbol; False / 0: off"""
        check(lib.vsr_sttn_timing(self._h, int(enable)))

    def timing_reset(self):
        check(lib.vsr_sttn_timing_reset(self._h))

    def timing_get(self, prefix=""):
        ms, n, fl = C.c_double(), C.c_int32(), C.c_double()
        check(lib.vsr_sttn_timing_get(self._h, prefix.encode(), C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value


def flow_timing(enable=True):
    """HIP events around every launch of the flow engines' plans (RAFT, flow completion, generator, LaMa); process-wide"""
    check(lib.vsr_flow_timing(1 if enable else 0))


def flow_timing_reset():
    check(lib.vsr_flow_timing_reset())


def flow_timing_get(prefix=""):
    """(ms, launches, algorithmic FLOPs) over the timed launches whose key starts with prefix (see include/vsr_hip.h)"""
    ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
    check(lib.vsr_flow_timing_get(prefix.encode(), C.byref(ms), C.byref(n), C.byref(fl)))
    return ms.value, n.value, fl.value


def flow_timing_keys():
    n = lib.vsr_flow_timing_keys(None, 0)
    buf = C.create_string_buffer(int(n))
    lib.vsr_flow_timing_keys(buf, n)
    return [k for k in buf.value.decode().split("\n") if k]


def flow_timing_by_kernel(engine):
    """{"gg:<tile cfg>:<bmode>:v<variant>": (ms, launches, flops)} of one engine ("raft", "rfc", "pp", "lama") + "op" for the rest"""
    out = {}
    for k in flow_timing_keys():
        parts = k.split(":")
        if parts[0] != engine:
            continue
        name = ":".join(parts[1:5]) if parts[1] == "gg" else "op"
        if name not in out:
            out[name] = flow_timing_get(f"{engine}:{name}:")
    return out


class RaftEngine:
    """RAFT ("things" configuration) resident on one GPU: the optical-flow stage of --inpaint-mode propainter
    (reference RAFT_bi, backend/inpaint/video/model/modules/flow_comp_raft.py:27-55)."""

    def __init__(self, state_dict, device=0):
        self._h = C.c_void_p()
        check(lib.vsr_raft_create(C.byref(self._h)))
        try:
            for key, val in state_dict.items():
                key = key[7:] if key.startswith("module.") else key      # DataParallel checkpoint (flow_comp_raft.py:17-19)
                arr = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
                arr = np.ascontiguousarray(arr, dtype=np.float32)
                shape = (C.c_int64 * arr.ndim)(*arr.shape)
                check(lib.vsr_raft_set_param(self._h, key.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim))
            if device is not None and device >= 0:
                require_gpu()
            self.device_index = -1 if device is None else int(device)
            check(lib.vsr_raft_finalize(self._h, self.device_index))
        except Exception:
            lib.vsr_raft_destroy(self._h)
            self._h = None
            raise
        self.device = torch.device("cuda", self.device_index) if self.device_index >= 0 else torch.device("cpu")

    def close(self):
        if getattr(self, "_h", None):
            lib.vsr_raft_destroy(self._h)
            self._h = None

    def set_precision(self, mode):
        """'f32' (default): exact fp32 contractions; 'split': fp16 hi/lo operand pairs with fp32 accumulation, range-guarded
        (a call that leaves the fp16 range is redone in fp32, see fallbacks())"""
        if mode not in FLOW_PRECISION_MODES:
            raise ValueError(f"precision {mode!r}: expected 'f32', 'split' or 'f16'")
        check(lib.vsr_raft_set_precision(self._h, FLOW_PRECISION_MODES[mode]))

    def fallbacks(self):
        return int(lib.vsr_raft_fallbacks(self._h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def packed_weights(self):
        n = lib.vsr_raft_packed_weights(self._h, None, 0)
        out = np.empty(n, dtype=np.float32)
        lib.vsr_raft_packed_weights(self._h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def flops(self, t, H, W, iters=20):
        return lib.vsr_raft_flops(self._h, t, H, W, iters)

    def read_buffer(self, buf, count, offset=0):
        """test hook: `count` floats of workspace buffer `buf` (plan buffer id) as a numpy array"""
        out = np.empty(count, dtype=np.float32)
        check(lib.vsr_raft_read_buffer(self._h, buf, offset, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def flows(self, frames_dev, iters=20, bgr=False):
        """frames_dev uint8 [t,H,W,3] on the GPU -> (forward, backward) flows, fp32 [t-1,2,H,W] each."""
        assert frames_dev.dtype == torch.uint8 and frames_dev.is_cuda and frames_dev.is_contiguous()
        t, H, W, _ = frames_dev.shape
        fwd = torch.empty((t - 1, 2, H, W), dtype=torch.float32, device=frames_dev.device)
        bwd = torch.empty_like(fwd)
        with torch.cuda.device(frames_dev.device):
            check(lib.vsr_raft_flows(self._h, C.c_void_p(frames_dev.data_ptr()), t, H, W, iters, 1 if bgr else 0,
                                     C.c_void_p(fwd.data_ptr()), C.c_void_p(bwd.data_ptr()), _stream_ptr()))
        return fwd, bwd


class RfcEngine:
    """Recurrent flow completion resident on one GPU: the second stage of --inpaint-mode propainter (reference
    RecurrentFlowCompleteNet.forward_bidirect_flow + combine_flow, recurrent_flow_completion.py:313-348)."""

    def __init__(self, state_dict, device=0):
        self._h = C.c_void_p()
        check(lib.vsr_rfc_create(C.byref(self._h)))
        try:
            for key, val in state_dict.items():
                arr = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
                arr = np.ascontiguousarray(arr, dtype=np.float32)
                shape = (C.c_int64 * arr.ndim)(*arr.shape)
                check(lib.vsr_rfc_set_param(self._h, key.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim))
            if device is not None and device >= 0:
                require_gpu()
            self.device_index = -1 if device is None else int(device)
            check(lib.vsr_rfc_finalize(self._h, self.device_index))
        except Exception:
            lib.vsr_rfc_destroy(self._h)
            self._h = None
            raise
        self.device = torch.device("cuda", self.device_index) if self.device_index >= 0 else torch.device("cpu")

    def close(self):
        if getattr(self, "_h", None):
            lib.vsr_rfc_destroy(self._h)
            self._h = None

    def set_precision(self, mode):
        """'f32' (default): exact fp32 contractions; 'split': fp16 hi/lo operand pairs with fp32 accumulation, range-guarded
        (a call that leaves the fp16 range is redone in fp32, see fallbacks())"""
        if mode not in FLOW_PRECISION_MODES:
            raise ValueError(f"precision {mode!r}: expected 'f32', 'split' or 'f16'")
        check(lib.vsr_rfc_set_precision(self._h, FLOW_PRECISION_MODES[mode]))

    def fallbacks(self):
        return int(lib.vsr_rfc_fallbacks(self._h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def packed_weights(self):
        n = lib.vsr_rfc_packed_weights(self._h, None, 0)
        out = np.empty(n, dtype=np.float32)
        lib.vsr_rfc_packed_weights(self._h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def flops(self, t, H, W):
        return lib.vsr_rfc_flops(self._h, t, H, W)

    def read_buffer(self, buf, count, offset=0):
        out = np.empty(count, dtype=np.float32)
        check(lib.vsr_rfc_read_buffer(self._h, buf, offset, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def complete(self, flows_f, flows_b, masks):
        """flows fp32 [t-1,2,H,W] and masks uint8 [t,H,W] (non-zero = hole) on the GPU -> completed (forward, backward) flows."""
        assert flows_f.dtype == torch.float32 and flows_f.is_cuda and flows_f.is_contiguous() and flows_b.is_contiguous()
        assert masks.dtype == torch.uint8 and masks.is_cuda and masks.is_contiguous()
        T, _, H, W = flows_f.shape
        assert masks.shape == (T + 1, H, W) and flows_b.shape == flows_f.shape
        of, ob = torch.empty_like(flows_f), torch.empty_like(flows_b)
        with torch.cuda.device(flows_f.device):
            check(lib.vsr_rfc_complete(self._h, C.c_void_p(flows_f.data_ptr()), C.c_void_p(flows_b.data_ptr()), C.c_void_p(masks.data_ptr()),
                                       T + 1, H, W, C.c_void_p(of.data_ptr()), C.c_void_p(ob.data_ptr()), _stream_ptr()))
        return of, ob


class PpEngine(AccuracyGuard):
    """ProPainter generator stages on one GPU (reference InpaintGenerator, backend/inpaint/video/model/propainter.py)."""

    def __init__(self, device=0, state_dict=None):
        """state_dict: torch.load('ProPainter.pth') -- needed by forward(), not by img_propagation()."""
        if device is not None and device >= 0:
            require_gpu()
        self.device_index = -1 if device is None else int(device)
        self._h = C.c_void_p()
        check(lib.vsr_pp_create(self.device_index, C.byref(self._h)))
        self.device = torch.device("cuda", self.device_index) if self.device_index >= 0 else torch.device("cpu")
        self._guard_init("f32", {"f16": "split", "split": "f32"})
        if state_dict is not None:
            try:
                for key, val in state_dict.items():
                    arr = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
                    arr = np.ascontiguousarray(arr, dtype=np.float32)
                    shape = (C.c_int64 * arr.ndim)(*arr.shape)
                    check(lib.vsr_pp_set_param(self._h, key.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim))
                check(lib.vsr_pp_finalize(self._h))
            except Exception:
                lib.vsr_pp_destroy(self._h)
                self._h = None
                raise

    @property
    def handle(self):
        return self._h

    def packed_weights(self):
        n = lib.vsr_pp_packed_weights(self._h, None, 0)
        out = np.empty(n, dtype=np.float32)
        lib.vsr_pp_packed_weights(self._h, out.ctypes.data_as(C.c_void_p), n)
        return out

    @staticmethod
    def window_flags(masks_local_host):
        """masks_in of the local frames, uint8 numpy [lt,H,W] -> one flag per attention window (numpy uint8)"""
        m = np.ascontiguousarray(masks_local_host, dtype=np.uint8)
        lt, H, W = m.shape
        flags = np.zeros(4096, dtype=np.uint8)
        n = lib.vsr_pp_window_flags(m.ctypes.data_as(C.c_void_p), lt, H, W, flags.ctypes.data_as(C.c_void_p), flags.size)
        if n < 0:
            check(n)
        return flags[:n].copy()

    def read_buffer(self, buf, count, offset=0):
        out = np.empty(count, dtype=np.float32)
        check(lib.vsr_pp_read_buffer(self._h, buf, offset, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def forward(self, frames, flows_f, flows_b, masks_in, masks_updated, lt, flags=None, box=None):
        """InpaintGenerator.forward in eval mode: frames fp32 [t,3,H,W], flows fp32 [lt-1,2,H,W], masks uint8 [t,H,W] on the GPU
        -> tanh output fp32 [lt,3,H,W].  box = (row_lo, row_hi, col_lo, col_hi): a promise that only that box of the output is read
        (vsr_pp_forward_box: the decoder runs on what the box depends on; outside it the output is undefined)."""
        return self._guarded_window(lambda: self._forward(frames, flows_f, flows_b, masks_in, masks_updated, lt, flags, box), box)

    def _guarded_window(self, raw, box):
        """one generator window under the accuracy guard (tanh output: a range of 2; inside the promised box only)"""
        if not self._guard_due():
            return raw()
        got = raw()
        mode = self._guard_exact()
        want = raw()
        g, w = (got, want) if not box or box[1] <= box[0] else (got[..., box[0]:box[1], box[2]:box[3]], want[..., box[0]:box[1], box[2]:box[3]])
        return got if self._guard_verdict(mode, g, w, None, 2.0) else want

    def _forward(self, frames, flows_f, flows_b, masks_in, masks_updated, lt, flags=None, box=None):
        assert frames.dtype == torch.float32 and frames.is_cuda and frames.is_contiguous()
        assert masks_in.dtype == torch.uint8 and masks_in.is_contiguous() and masks_updated.is_contiguous()
        t, _, H, W = frames.shape
        if flags is None:
            flags = self.window_flags(masks_in[:lt].cpu().numpy())
        out = torch.empty((lt, 3, H, W), dtype=torch.float32, device=frames.device)
        with torch.cuda.device(frames.device):
            check(lib.vsr_pp_forward_box(self._h, C.c_void_p(frames.data_ptr()), C.c_void_p(flows_f.data_ptr()), C.c_void_p(flows_b.data_ptr()),
                                         C.c_void_p(masks_in.data_ptr()), C.c_void_p(masks_updated.data_ptr()), t, lt, H, W,
                                         flags.ctypes.data_as(C.c_void_p), flags.size, *[int(b) for b in (box or (0, 0, 0, 0))],
                                         C.c_void_p(out.data_ptr()), _stream_ptr()))
        return out

    def encode(self, frames, masks_in, masks_updated, ntok_frames=0):
        """The generator's encoder (and, for the first ntok_frames frames, the soft split) once per frame: frames fp32 [n,3,H,W], masks
        uint8 [n,H,W] on the GPU -> (features fp32 [n,H/4,W/4,128], tokens fp32 [ntok_frames,tokens,512]) for forward_cached."""
        assert frames.dtype == torch.float32 and frames.is_cuda and frames.is_contiguous()
        assert masks_in.dtype == torch.uint8 and masks_in.is_contiguous() and masks_updated.is_contiguous()
        n, _, H, W = frames.shape
        feats = torch.empty((n, H // 4, W // 4, 128), dtype=torch.float32, device=frames.device)
        toks = torch.empty((ntok_frames, int(lib.vsr_pp_token_count(H, W)), 512), dtype=torch.float32, device=frames.device)
        with torch.cuda.device(frames.device):
            check(lib.vsr_pp_encode(self._h, C.c_void_p(frames.data_ptr()), C.c_void_p(masks_in.data_ptr()), C.c_void_p(masks_updated.data_ptr()),
                                    n, int(ntok_frames), H, W, C.c_void_p(feats.data_ptr()),
                                    C.c_void_p(toks.data_ptr()) if ntok_frames else None, _stream_ptr()))
        return feats, toks

    def forward_cached(self, feat_cache, tok_cache, cache_idx, flows_f, flows_b, masks_in, masks_updated, lt, H, W, flags, box=None):
        """forward() from cached per-frame encoder output: cache_idx[k] = entry of feat_cache for the local frame k < lt, entry of
        tok_cache for the reference frame k >= lt; masks uint8 [t,H,W] of the window's frames."""
        return self._guarded_window(lambda: self._forward_cached(feat_cache, tok_cache, cache_idx, flows_f, flows_b, masks_in, masks_updated, lt, H, W,
                                                                 flags, box), box)

    def _forward_cached(self, feat_cache, tok_cache, cache_idx, flows_f, flows_b, masks_in, masks_updated, lt, H, W, flags, box=None):
        assert feat_cache.dtype == torch.float32 and feat_cache.is_cuda and feat_cache.is_contiguous()
        assert tok_cache is None or (tok_cache.dtype == torch.float32 and tok_cache.is_contiguous())
        assert masks_in.dtype == torch.uint8 and masks_in.is_contiguous() and masks_updated.is_contiguous()
        idx = np.ascontiguousarray(np.asarray(cache_idx, dtype=np.int32))
        t = int(idx.size)
        assert tuple(masks_in.shape) == (t, H, W) and 1 <= lt <= t
        assert int(idx[:lt].max()) < feat_cache.shape[0] and (t == lt or int(idx[lt:].max()) < tok_cache.shape[0])
        out = torch.empty((lt, 3, H, W), dtype=torch.float32, device=feat_cache.device)
        with torch.cuda.device(feat_cache.device):
            check(lib.vsr_pp_forward_cached(self._h, C.c_void_p(feat_cache.data_ptr()),
                                            C.c_void_p(tok_cache.data_ptr()) if tok_cache is not None and tok_cache.numel() else None,
                                            idx.ctypes.data_as(C.c_void_p), C.c_void_p(flows_f.data_ptr()), C.c_void_p(flows_b.data_ptr()),
                                            C.c_void_p(masks_in.data_ptr()), C.c_void_p(masks_updated.data_ptr()), t, lt, H, W,
                                            flags.ctypes.data_as(C.c_void_p), flags.size, *[int(b) for b in (box or (0, 0, 0, 0))],
                                            C.c_void_p(out.data_ptr()), _stream_ptr()))
        return out

    def plan_flops(self, t, lt, H, W, flags, box=None, mode=0):
        """algorithmic FLOPs (2*M*N*K of every contraction) of one generator plan: mode 0 = forward(), 1 = encode() of t frames,
        2 = forward_cached(); the plan is built on the host and dropped again (bench / profile use only)"""
        f = np.ascontiguousarray(flags, dtype=np.uint8)
        plan = C.c_void_p()
        check(lib.vsr_pp_gen_plan_create_mode(self._h, int(t), int(lt), int(H), int(W), f.ctypes.data_as(C.c_void_p) if f.size else None, f.size,
                                              *[int(b) for b in (box or (0, 0, 0, 0))], int(mode), C.byref(plan)))
        try:
            return float(lib.vsr_plan_flops(plan))
        finally:
            lib.vsr_plan_destroy(plan)

    def close(self):
        if getattr(self, "_h", None):
            lib.vsr_pp_destroy(self._h)
            self._h = None

    def _apply_precision(self, mode):
        """'f32' (default): exact fp32 contractions; 'split': fp16 hi/lo operand pairs with fp32 accumulation, range-guarded
        (a call that leaves the fp16 range is redone in fp32, see fallbacks()); 'f16': fp16 operands, fp32 accumulation, range-guarded.
        (set_precision = this + the accuracy guard's first-unit check, AccuracyGuard)"""
        if mode not in FLOW_PRECISION_MODES:
            raise ValueError(f"precision {mode!r}: expected 'f32', 'split' or 'f16'")
        check(lib.vsr_pp_set_precision(self._h, FLOW_PRECISION_MODES[mode]))
        self.precision = mode

    def fallbacks(self):
        """calls redone in exact fp32: the kernels' range guard + the accuracy guard's demotions (AccuracyGuard)"""
        return int(lib.vsr_pp_fallbacks(self._h)) + self.demotions

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def img_propagation(self, masked_frames, flows_f, flows_b, masks):
        """InpaintGenerator.img_propagation(..., 'nearest'): masked_frames fp32 [t,3,H,W], flows fp32 [t-1,2,H,W], masks uint8
        [t,H,W] on the GPU -> (propagated frames fp32 [t,3,H,W], updated masks uint8 [t,H,W])."""
        assert masked_frames.dtype == torch.float32 and masked_frames.is_cuda and masked_frames.is_contiguous()
        assert masks.dtype == torch.uint8 and masks.is_contiguous() and flows_f.is_contiguous() and flows_b.is_contiguous()
        t, _, H, W = masked_frames.shape
        out = torch.empty_like(masked_frames)
        om = torch.empty_like(masks)
        with torch.cuda.device(masked_frames.device):
            check(lib.vsr_pp_img_propagation(self._h, C.c_void_p(masked_frames.data_ptr()), C.c_void_p(flows_f.data_ptr()),
                                             C.c_void_p(flows_b.data_ptr()), C.c_void_p(masks.data_ptr()), t, H, W,
                                             C.c_void_p(out.data_ptr()), C.c_void_p(om.data_ptr()), _stream_ptr()))
        return out, om


class LamaEngine:
    """big-LaMa generator resident on one GPU (reference: the TorchScript module of backend/inpaint/lama_inpaint.py:13 and the
    array work around its call, :45-60).  state_dict: the generator's entries (`model.N...`, optional `generator.` prefix)."""

    def __init__(self, state_dict, device=0):
        self._h = C.c_void_p()
        check(lib.vsr_lama_create(C.byref(self._h)))
        try:
            for key, val in state_dict.items():
                if key.endswith("num_batches_tracked"):
                    continue
                arr = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
                arr = np.ascontiguousarray(arr, dtype=np.float32)
                shape = (C.c_int64 * arr.ndim)(*arr.shape)
                check(lib.vsr_lama_set_param(self._h, key.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim))
            if device is not None and device >= 0:
                require_gpu()
            self.device_index = -1 if device is None else int(device)
            check(lib.vsr_lama_finalize(self._h, self.device_index))
        except Exception:
            lib.vsr_lama_destroy(self._h)
            self._h = None
            raise
        self.device = torch.device("cuda", self.device_index) if self.device_index >= 0 else torch.device("cpu")
        self.n_blocks = int(lib.vsr_lama_blocks(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib.vsr_lama_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def set_precision(self, mode):
        if mode not in FLOW_PRECISION_MODES:
            raise ValueError(f"precision {mode!r}: expected 'f32', 'split' or 'f16'")
        check(lib.vsr_lama_set_precision(self._h, FLOW_PRECISION_MODES[mode]))

    def fallbacks(self):
        return int(lib.vsr_lama_fallbacks(self._h))

    def packed_weights(self):
        n = lib.vsr_lama_packed_weights(self._h, None, 0)
        out = np.empty(n, dtype=np.float32)
        lib.vsr_lama_packed_weights(self._h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def flops(self, B, H, W):
        return lib.vsr_lama_flops(self._h, B, H, W)

    def read_buffer(self, buf, count, offset=0):
        out = np.empty(count, dtype=np.float32)
        check(lib.vsr_lama_read_buffer(self._h, buf, offset, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def inpaint(self, images, mask, out=None):
        """images uint8 [B,H,W,3] on the GPU (rows contiguous; a row-slice of a frame batch is fine), mask uint8 [H,W] (one for
        all) or [B,H,W] (non-zero = hole) -> uint8 [B,H,W,3]; out=None allocates, out=images works in place."""
        assert images.dtype == torch.uint8 and images.is_cuda and images.dim() == 4 and images.shape[3] == 3
        B, H, W, _ = images.shape
        assert images.stride(3) == 1 and images.stride(2) == 3 and images.stride(1) == 3 * W, "image rows must be contiguous"
        assert mask.dtype == torch.uint8 and mask.is_cuda
        if mask.dim() == 2:
            m, mstride = mask.contiguous(), 0
            assert tuple(m.shape) == (H, W)
        else:
            m = mask
            assert tuple(m.shape) == (B, H, W) and m.stride(2) == 1 and m.stride(1) == W
            mstride = m.stride(0) if B > 1 else H * W
        if out is None:
            out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=images.device)
        assert out.stride(3) == 1 and out.stride(2) == 3 and out.stride(1) == 3 * W
        istr = images.stride(0) if B > 1 else H * W * 3
        ostr = out.stride(0) if B > 1 else H * W * 3
        with torch.cuda.device(images.device):
            check(lib.vsr_lama_inpaint(self._h, C.c_void_p(images.data_ptr()), istr, C.c_void_p(m.data_ptr()), mstride, B, H, W,
                                       C.c_void_p(out.data_ptr()), ostr, _stream_ptr()))
        return out
