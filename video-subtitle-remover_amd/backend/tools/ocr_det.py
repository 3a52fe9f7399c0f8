"""Text detector on the MI355X: the `text_detector` object the reference builds in backend/tools/subtitle_detect.py:41-54
(paddleocr.TextDetection over backend/models/V5/{ch_det,ch_det_fast}: PP-OCRv5 server / mobile detection) -- SURVEY.md 8(a) a20.

  TextDetection(model_dir | graph, weights, device).predict(img) -> [{"dt_polys": int32 [n,4,2], "dt_scores": [n]}]

Pre-processing follows the model's inference.yml (DetResizeForTest resize_long 960, NormalizeImage ImageNet mean / std on the
BGR image, CHW), the forward pass executes the PaddlePaddle inference program (inference.json) operator by operator with the
HIP kernels of csrc/det_kernels.hip (walked here, on the host, the way Paddle's executor walks it), post-processing is
DBPostProcess (thresh 0.3, box_thresh 0.6, max_candidates 1000, unclip_ratio 1.5): threshold, connected components and their
bounding boxes on the device (DeviceDBPostProcess), the polygon geometry of the few components on the host.
There is no CPU path: the runner needs a HIP device.  paddleocr / paddlepaddle and the *.pdiparams weights are absent from the
reference mount, so `weights` is a {parameter name: array} dict (tests use synthetic ones) -- parity with Paddle's binary
is unpinned.
"""
import ctypes as C
import os

import numpy as np
import scipy.ndimage
import torch

from ... import _lib
from ..._lib import check, lib
from . import ocr_det_nhwc
from .paddle_graph import load_graph, read_pdiparams


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


TILE_128x128, TILE_128x64, BMODE_NK = 0, 3, 0      # include/vsr_hip.h: VSR_TILE_*, VSR_BMODE_NK


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


GEMM_MIN_K = 64          # dense convs with cin * kh * kw below this stay on the direct kernel (the 3-channel stem)


def conv_gemm_layout(cin, h, w, wshape, strides, pt, pl, ho, wo, n=1):
    """Host-side description of one dense convolution as a gather-GEMM problem (include/vsr_hip.h GGProblem, NK mode):
    A = the n input images as zero-padded NHWC [n][Hp][Wp][Cp] (image origin at (pt, pl), Cp = cin rounded up to the 32-float K
    chunk), rows = output pixels of all images, K = (ky, kx, channel); B = the weights re-packed [cout][kh*kw*Cp]; C = [n * pixels][Ncs]
    row-major.  Returns the dims and the int32 offset tables (padded to whole tiles as the kernels expect)."""
    cout, wcin, kh, kw = wshape
    assert wcin == cin
    sh, sw = strides
    cp = -(-cin // 32) * 32
    hp, wp = max(h + pt, (ho - 1) * sh + kh), max(w + pl, (wo - 1) * sw + kw)
    P = ho * wo
    K, M, N = kh * kw * cp, n * P, cout
    bm, bn, cfg = (128, 128, TILE_128x128) if cout >= 128 else (128, 64, TILE_128x64)
    tiles_m, tiles_n = -(-M // bm), -(-N // bn)
    ncs = tiles_n * bn
    if n * hp * wp * cp >= 2 ** 31 or M * ncs >= 2 ** 31 or N * K >= 2 ** 31:
        raise ValueError("convolution too large for 32-bit offset tables")
    img, pix = np.divmod(np.arange(M, dtype=np.int64), P)
    oy, ox = np.divmod(pix, wo)
    row_a = np.zeros(tiles_m * bm, np.int64)
    row_a[:M] = ((img * hp + oy * sh) * wp + ox * sw) * cp
    row_a[M:] = row_a[0]
    tap, cc = np.divmod(np.arange(K // 32, dtype=np.int64), cp // 32)
    col_a = ((tap // kw) * wp + tap % kw) * cp + cc * 32
    row_b = np.zeros(tiles_n * bn, np.int64)
    row_b[:N] = np.arange(N, dtype=np.int64) * K
    col_b = np.arange(K // 32, dtype=np.int64) * 32
    row_c = np.zeros(tiles_m * bm, np.int64)
    row_c[:M] = np.arange(M, dtype=np.int64) * ncs
    col_c = np.arange(ncs // 32, dtype=np.int64) * 32
    t = {k: v.astype(np.int32) for k, v in dict(rowA=row_a, colA=col_a, rowB=row_b, colB=col_b, rowC=row_c, colC=col_c).items()}
    return dict(cp=cp, hp=hp, wp=wp, K=K, M=M, N=N, P=P, n=n, ncs=ncs, tile_cfg=cfg, tiles_m=tiles_m, tiles_n=tiles_n, tables=t)


def conv_fusions(graph):
    """Per conv2d op, what can be folded into the pass that writes its output: {conv op index: (affine, act)} with
    affine = None | ("bn", op index) | ("bias", op index, id of the bias parameter)   -- a batch_norm_ or the add of a reshaped
    parameter reading the conv's result, act = None | (op index, 1 relu / 2 hardswish) reading that; every folded intermediate
    has exactly one reader."""
    uses, consumer, producer = {}, {}, {}
    for i, (kind, ins, outs, a) in enumerate(graph.ops):
        for v in ins:
            uses[v] = uses.get(v, 0) + 1
            consumer.setdefault(v, []).append(i)
        for v in outs:
            producer[v] = i
    uses[graph.output_id] = uses.get(graph.output_id, 0) + 1

    def sole_reader(v):
        c = consumer.get(v, [])
        return c[0] if uses.get(v, 0) == 1 and len(c) == 1 else None

    def param_behind(v):                      # the parameter a bias operand is (a reshape of), else None
        if v in graph.params:
            return v
        j = producer.get(v)
        return graph.ops[j][1][0] if j is not None and graph.ops[j][0] == "reshape" and graph.ops[j][1][0] in graph.params else None

    fuse = {}
    for i, (kind, ins, outs, a) in enumerate(graph.ops):
        if kind != "conv2d":
            continue
        affine = act = None
        j = sole_reader(outs[0])
        if j is not None:
            k2, i2, o2, _ = graph.ops[j]
            if k2 == "batch_norm_" and i2[0] == outs[0]:
                affine = ("bn", j)
            elif k2 == "add" and len(i2) == 2 and param_behind(i2[1] if i2[0] == outs[0] else i2[0]) is not None:
                affine = ("bias", j, param_behind(i2[1] if i2[0] == outs[0] else i2[0]))
            if affine is not None:
                r = sole_reader(o2[0])
                if r is not None and graph.ops[r][0] in ("relu", "hardswish"):
                    act = (r, 1 if graph.ops[r][0] == "relu" else 2)
        fuse[i] = (affine, act)
    return fuse


def deconv_fusions(graph):
    """Per conv2d_transpose op, the chain that can ride on its GEMM form (PaddleGraphRunner._deconv_gemm): {op index: (bias, bn, act)}
    with bias = None | (add op index, id of the bias parameter) -- folded into the GEMM's bias --, bn = None | batch_norm_ op index and
    act = None | (op index, 1 relu / 2 hardswish) -- both applied by the pass that writes the NCHW result; every folded intermediate
    has exactly one reader."""
    uses, consumer, producer = {}, {}, {}
    for i, (kind, ins, outs, a) in enumerate(graph.ops):
        for v in ins:
            uses[v] = uses.get(v, 0) + 1
            consumer.setdefault(v, []).append(i)
        for v in outs:
            producer[v] = i
    uses[graph.output_id] = uses.get(graph.output_id, 0) + 1

    def sole_reader(v):
        c = consumer.get(v, [])
        return c[0] if uses.get(v, 0) == 1 and len(c) == 1 else None

    def param_behind(v):
        if v in graph.params:
            return v
        j = producer.get(v)
        return graph.ops[j][1][0] if j is not None and graph.ops[j][0] == "reshape" and graph.ops[j][1][0] in graph.params else None

    fuse = {}
    for i, (kind, ins, outs, a) in enumerate(graph.ops):
        if kind != "conv2d_transpose":
            continue
        bias = bn = act = None
        cur = outs[0]
        j = sole_reader(cur)
        if j is not None and graph.ops[j][0] == "add" and len(graph.ops[j][1]) == 2:
            i2 = graph.ops[j][1]
            pb = param_behind(i2[1] if i2[0] == cur else i2[0])
            if pb is not None:
                bias, cur = (j, pb), graph.ops[j][2][0]
                j = sole_reader(cur)
        if j is not None and graph.ops[j][0] == "batch_norm_" and graph.ops[j][1][0] == cur:
            bn, cur = j, graph.ops[j][2][0]
            j = sole_reader(cur)
        if (bias is not None or bn is not None) and j is not None and graph.ops[j][0] in ("relu", "hardswish"):
            act = (j, 1 if graph.ops[j][0] == "relu" else 2)
        fuse[i] = (bias, bn, act)
    return fuse


def pack_conv_weights(w, cp):
    """[cout][cin][kh][kw] -> [cout][(ky, kx, channel padded to cp)], the K order of conv_gemm_layout"""
    cout, cin, kh, kw = w.shape
    out = np.zeros((cout, kh * kw, cp), np.float32)
    out[:, :, :cin] = np.asarray(w, np.float32).transpose(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    return out.reshape(cout, kh * kw * cp)


class PaddleGraphRunner:
    """Executes a detection program on one GPU; values are NCHW fp32 torch tensors (device memory only).  Dense convolutions run
    on the gather-GEMM (exact fp32 MFMA) between two layout kernels, with the batch_norm (+ ReLU) that follows folded into the
    second one; depthwise and the 3-channel stem stay on the direct kernel."""

    def __init__(self, graph, weights, device=0):
        if lib.vsr_device_count() <= 0 or not torch.cuda.is_available():
            raise _lib.VsrError(_lib.VSR_ERR_NOGPU, "no HIP device visible: the text detector has no CPU fallback")
        self.graph = graph
        self.device = torch.device("cuda", device)
        self.params = {}
        for vid, (name, shape) in graph.params.items():
            if name not in weights:
                raise KeyError(f"missing detector parameter: {name}")
            a = np.ascontiguousarray(np.asarray(weights[name], dtype=np.float32))
            if tuple(a.shape) != tuple(shape):
                raise ValueError(f"shape mismatch for {name}: {a.shape} vs {shape}")
            self.params[vid] = torch.from_numpy(a).to(self.device)
        # inference batch_norm as a per-channel affine (inputs: x, mean, variance, scale, bias)
        self.bn = {}
        for i, (kind, ins, outs, a) in enumerate(graph.ops):
            if kind == "batch_norm_":
                mean, var, gamma, beta = (self.params[j].double() for j in ins[1:5])
                s = gamma / torch.sqrt(var + a["epsilon"])
                self.bn[i] = (s.float().contiguous(), (beta - mean * s).float().contiguous())
        self._fuse = conv_fusions(graph)      # conv op index -> (bn op index or None, relu op index or None)
        self._dfuse = deconv_fusions(graph)   # conv2d_transpose op index -> (bias, bn, act) riding on its GEMM form
        self._one = {}
        self._graphs = {}                     # input shape -> (captured graph, static input, static output)
        self._gemm = {}                       # (conv op index, input shape) -> resident plan, tables, buffers
        self.use_gemm = os.environ.get("VSR_DET_GEMM", "1") != "0"
        self.deconv_gemm = os.environ.get("VSR_DET_DECONV_GEMM", "1") != "0"     # 0: the 2x2 transposed convs stay on the direct kernel
        self._sa = C.c_void_p(0)              # the stream argument every launch shares (set per run / replay)
        self._tape = None                     # launches being recorded: [(C function, argument tuple)]
        self._tapes = {}                      # input shape -> (tape, static input, output, tensors kept alive)
        self.flops = {}                       # input shape -> algorithmic FLOPs of one forward (2 * MACs of every conv / transposed conv)
        # NHWC-resident compiled plan (ocr_det_nhwc.py): "1" always, "0" never (the recorded op-by-op walk), "auto" (default) when the
        # program compiles to few layout passes (the server program: 3 for 114 GEMMs; the mobile program's squeeze-excite / hardswish
        # blocks keep it on the NCHW kernels)
        self.nhwc = os.environ.get("VSR_DET_NHWC", "auto")
        self._plans = {}                      # input shape -> instantiated plan (buffers, constants, resident GEMM plans, launch list) or None

    def _c(self, t):
        """t as a contiguous tensor.  While a tape is open an implicit torch copy would be a launch the tape does not hold (and
        whose result nothing keeps alive): the replay would read stale memory.  No op of the two shipped programs needs one;
        a program that does is refused instead of replayed wrongly."""
        if t.is_contiguous():
            return t
        if self._tape is not None:
            raise NotImplementedError("detector tape: an operand is not contiguous -- its copy would not be recorded; "
                                      "run with VSR_DET_TAPE=0 (op-by-op walk) for this program")
        return t.contiguous()

    def _call(self, fn, *args):
        """one launcher call; recorded when a tape is open"""
        check(fn(*args))
        if self._tape is not None:
            self._tape.append((fn, args))

    def plan_for(self, shape):
        """the instantiated NHWC plan of this input shape, or None when the program does not lend itself to one (VSR_DET_NHWC)"""
        key = tuple(int(v) for v in shape)
        if key in self._plans:
            return self._plans[key]
        st = None
        if self.nhwc != "0" and self.use_gemm:
            params = {vid: t.cpu().numpy() for vid, t in self.params.items()}
            try:
                plan = ocr_det_nhwc.compile_plan(self.graph, params, key)
            except ValueError:                      # a buffer beyond the 32-bit offset tables: the walk (which checks conv by conv) decides
                plan = None
            except NotImplementedError:             # an operator form the compiler has no NHWC / NCHW step for: "auto" leaves the program to
                if self.nhwc == "1":                # the op-by-op walk (the same HIP kernels, operator by operator), "1" insists
                    raise
                plan = None
            kinds = {}
            for k, _ in (plan.steps if plan is not None else []):
                kinds[k] = kinds.get(k, 0) + 1
            passes = kinds.get("to_view", 0) + kinds.get("from_view", 0)
            if plan is not None and (self.nhwc == "1" or passes * 4 <= kinds.get("gemm", 0)):
                with torch.cuda.device(self.device):
                    st = self._instantiate(plan)
                st["kinds"] = kinds
                self.flops[key] = plan.flops
        self._plans[key] = st
        return st

    def _instantiate(self, plan):
        """buffers, constants and resident GEMM plans of a compiled plan on the device + its launch list [(C function, arguments)]"""
        dev = self.device
        bufs = {k: (torch.zeros if zero else torch.empty)(max(sz, 4), dtype=torch.float32, device=dev) for k, (sz, zero) in plan.buffers.items()}
        consts = {k: torch.from_numpy(a).to(dev) for k, a in plan.consts.items()}
        bp = lambda name, off=0: C.c_void_p(bufs[name].data_ptr() + 4 * int(off))
        cp = lambda name: C.c_void_p(consts[name].data_ptr()) if name is not None else None
        sa = self._sa
        tape, handles, pending = [], [], []

        def flush():
            """the GEMM steps gathered so far (one, or a group that shares nothing) as one resident launch list"""
            if not pending:
                return
            pr = (_lib.GGProblem * len(pending))()
            for q, p in zip(pr, pending):
                t = p["tables"]
                q.A, q.B, q.C = bufs[p["A"]].data_ptr(), consts[p["B"]].data_ptr(), bufs[p["C"]].data_ptr()
                q.bias = consts[p["bias"]].data_ptr() if p["bias"] is not None else None
                q.R = bufs[p["R"]].data_ptr() if p["R"] is not None else None
                q.rowA, q.colA, q.rowB, q.colB = (consts[t[k]].data_ptr() for k in ("rowA", "colA", "rowB", "colB"))
                q.rowC, q.colC = consts[t["rowC"]].data_ptr(), consts[t["colC"]].data_ptr()
                q.rowR = consts[t["rowR"]].data_ptr() if "rowR" in t else None
                q.M, q.N, q.K, q.tilesM, q.tilesN = p["M"], p["N"], p["K"], p["tiles_m"], p["tiles_n"]
                q.splitK, q.chunksPerSplit, q.act, q.alpha, q.splitStride = 1, p["K"] // 32, p["act"], 1.0, 0
            h = C.c_void_p()
            check(lib.vsr_gemm_plan_create(pr, len(pending), pending[0]["tile_cfg"], BMODE_NK, pending[0]["variant"], C.byref(h)))
            handles.append(h)
            tape.append((lib.vsr_gemm_plan_run, (h, sa)))
            launches.append(("gemm", list(pending)))
            pending.clear()

        launches = []                             # what each tape entry is: (kind, step parameters) -- for per-launch timing
        for kind, p in plan.steps:
            if kind == "gemm":
                if p["group"] is not None and pending and pending[0]["group"] == p["group"]:
                    pending.append(p)
                    continue
                flush()
                pending.append(p)
                continue
            flush()
            if kind == "to_view":
                tape.append((lib.vsr_det_launch_to_view, (bp(p["x"]), p["n"], p["C"], p["H"], p["W"], p["Cw"], bp(p["out"], p["out_off"]), p["img_stride"],
                                                           p["row_stride"], p["Cs"], sa)))
            elif kind == "from_view":
                tape.append((lib.vsr_det_launch_from_view, (bp(p["inp"], p["in_off"]), p["img_stride"], p["row_stride"], p["Cs"], p["n"], p["C"], p["H"], p["W"],
                                                             bp(p["out"], p["out_off"]), p["out_img_stride"], sa)))
            elif kind == "dwconv_view":
                tape.append((lib.vsr_det_launch_dwconv_view, (bp(p["inp"], p["in_off"]), p["in_img"], p["in_row"], p["in_cs"], cp(p["w"]), cp(p["scale"]), cp(p["shift"]),
                                                               p["n"], p["C"], p["kh"], p["kw"], p["sh"], p["sw"], p["pt"], p["pl"], p["Ho"], p["Wo"], p["act"],
                                                               bp(p["out"], p["out_off"]), p["out_img"], p["out_row"], p["out_cs"], sa)))
            elif kind == "nearest_view":
                tape.append((lib.vsr_det_launch_nearest_view, (bp(p["inp"], p["in_off"]), p["in_img"], p["in_row"], p["in_cs"], p["n"], p["C"], p["Ho"], p["Wo"], p["s"],
                                                                bp(p["out"], p["out_off"]), p["out_img"], p["out_row"], p["out_cs"], sa)))
            elif kind == "im2col_view":
                tape.append((lib.vsr_det_launch_im2col_view, (bp(p["x"]), p["n"], p["C"], p["H"], p["W"], p["kh"], p["kw"], p["pt"], p["pl"], bp(p["out"], p["out_off"]),
                                                               p["out_img"], p["out_row"], p["out_cs"], sa)))
            elif kind == "dots_view":
                tape.append((lib.vsr_det_launch_dots_view, (bp(p["inp"], p["in_off"]), p["in_img"], p["in_row"], p["in_cs"], p["n"], p["C"], p["H"], p["W"], cp(p["w"]),
                                                             cp(p["bias"]), p["n_out"], p["act"], bp(p["out"]), sa)))
            elif kind == "conv_nchw":
                tape.append((lib.vsr_det_launch_conv2d, (bp(p["x"]), cp(p["w"]), None, p["n"], p["cin"], p["h"], p["wd"], p["cout"], p["kh"], p["kw"], p["sh"], p["sw"],
                                                          p["pt"], p["pl"], p["ho"], p["wo"], p["dw"], 0, bp(p["out"]), sa)))
            elif kind == "deconv_nchw":
                tape.append((lib.vsr_det_launch_deconv2x2, (bp(p["x"]), cp(p["w"]), p["n"], p["cin"], p["h"], p["wd"], p["cout"], p["dw"], bp(p["out"]), sa)))
            elif kind == "affine":
                tape.append((lib.vsr_det_launch_affine, (bp(p["x"]), cp(p["scale"]), cp(p["shift"]), p["total"], p["C"], p["HW"], bp(p["out"]), sa)))
            elif kind == "binary":
                b = cp(p["b"][1]) if p["b"][0] == "const" else bp(p["b"][1])
                tape.append((lib.vsr_det_launch_binary, (bp(p["a"]), b, p["op"], p["total"], p["C"], p["HW"], p["mode"], bp(p["out"]), sa)))
            elif kind == "unary":
                tape.append((lib.vsr_det_launch_unary, (bp(p["x"]), p["total"], p["kind"], C.c_float(p["p0"]), C.c_float(p["p1"]), bp(p["out"]), sa)))
            elif kind == "gap":
                tape.append((lib.vsr_det_launch_gap, (bp(p["x"]), p["planes"], p["HW"], bp(p["out"]), sa)))
            elif kind == "maxpool":
                tape.append((lib.vsr_det_launch_maxpool, (bp(p["x"]), p["planes"], p["H"], p["W"], p["kh"], p["kw"], p["sh"], p["sw"], p["pt"], p["pl"], p["Ho"],
                                                           p["Wo"], bp(p["out"]), sa)))
            elif kind == "nearest_nchw":
                tape.append((lib.vsr_det_launch_nearest, (bp(p["x"]), p["planes"], p["H"], p["W"], p["s"], bp(p["out"]), sa)))
            elif kind == "copy":
                tape.append((lib.vsr_det_launch_copy, (bp(p["src"]), 4 * p["src_pitch"], bp(p["dst"], p["dst_off"]), 4 * p["dst_pitch"], 4 * p["width"], p["rows"], sa)))
            else:
                raise NotImplementedError(kind)
            launches.append((kind, p))
        flush()
        name, shape = plan.output
        out = bufs[name][:int(np.prod(shape))].view(*shape)
        xin = bufs[plan.input][:plan.buffers[plan.input][0]]
        return dict(tape=tape, bufs=bufs, consts=consts, gemm_plans=handles, out=out, xin=xin, plan=plan, launches=launches)

    def run_planned(self, x, st=None):
        """the forward from the compiled NHWC plan of x's shape (plan_for): the input is copied into the plan's buffer, the launch list is
        issued on the caller's current stream; the returned tensor is the plan's output buffer, valid until the next call for that shape"""
        st = st if st is not None else self.plan_for(x.shape)
        if st is None:
            raise _lib.VsrError(_lib.VSR_ERR_ARG, "no NHWC plan for this program (VSR_DET_NHWC)")
        with torch.cuda.device(self.device):
            st["xin"].copy_(x.reshape(-1))
            self._sa.value = torch.cuda.current_stream().cuda_stream
            for fn, args in st["tape"]:
                rc = fn(*args)
                if rc != 0:
                    check(rc)
        return st["out"]

    def run_taped(self, x):
        """run() replayed from a recorded launch list.  The forward is 300-450 launches and, walked op by op in Python (shape
        logic, allocations, ctypes marshalling), host-bound; every launch of a fixed input shape has fixed arguments, so the
        second pass for a shape records (C function, arguments) with all intermediates kept alive and later passes only copy the
        input into the recorded input buffer and issue the recorded calls.  Unlike the HIP-graph replay (run_graphed) nothing is
        captured by the driver: the same launches, on the caller's current stream.  The returned tensor is the recorded output
        buffer: valid until the next call for that shape."""
        key = tuple(x.shape)
        pst = self.plan_for(key)
        if pst is not None:
            return self.run_planned(x, pst)
        st = self._tapes.get(key)
        if st is None:
            with torch.cuda.device(self.device):
                self.run(x)                                     # creates the resident GEMM plans of this shape
                sx = x.clone()
                self._tape, self._keep = [], []
                try:
                    out = self.run(sx)
                    st = (self._tape, sx, out, self._keep)
                finally:
                    self._tape, self._keep = None, None
            self._tapes[key] = st
            return out
        tape, sx, out, _ = st
        with torch.cuda.device(self.device):
            sx.copy_(x)
            self._sa.value = torch.cuda.current_stream().cuda_stream
            for fn, args in tape:
                rc = fn(*args)
                if rc != 0:
                    check(rc)
        return out

    def close(self):
        self._tapes.clear()
        self._graphs.clear()                  # captured graphs reference the plans' buffers: drop them first
        for st in self._plans.values():
            if st is not None:
                for h in st["gemm_plans"]:
                    lib.vsr_gemm_plan_destroy(h)
        self._plans.clear()
        for st in self._gemm.values():
            lib.vsr_gemm_plan_destroy(st["plan"])
        self._gemm.clear()

    def run_graphed(self, x):
        """Opt-in (VSR_DET_GRAPH=1).  Round 1's fault at replay: captured hipMemsetAsync nodes (vsr_gemm_plan_run zeroed its tile counters
        with one per conv) -- a graph holding them faulted on a later replay in 4 of 9 processes, a graph of kernel nodes only in 0 of 9
        (profiles/r06_det_graph_triage.log); the plans zero with a kernel since.
        run() replayed from a HIP graph: the forward is ~300-450 small launches and host-bound when issued one by one.  The first
        call for an input shape runs eagerly once (creates the resident GEMM plans: allocations and uploads cannot be captured), then
        captures a second pass; later calls copy the input into the captured buffer and replay.  The returned tensor is the graph's
        output buffer: valid until the next call."""
        key = tuple(x.shape)
        st = self._graphs.get(key)
        if st is None:
            with torch.cuda.device(self.device):
                self.run(x)
                torch.cuda.synchronize(self.device)
                sx = x.clone()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    out = self.run(sx)
            st = (gr, sx, out)
            self._graphs[key] = st
        gr, sx, out = st
        sx.copy_(x)
        gr.replay()
        return out

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _conv_gemm(self, i, xin, w, strides, pt, pl, ho, wo):
        """one dense conv (any batch) through the gather-GEMM; returns (output NCHW, set of op indices folded into it)"""
        nb, cin, h, wd = xin.shape
        key = (i, tuple(xin.shape))
        st = self._gemm.get(key)
        if st is None:
            lay = conv_gemm_layout(cin, h, wd, tuple(w.shape), strides, pt, pl, ho, wo, n=nb)
            dev = self.device
            st = dict(lay=lay, tabs={k: torch.from_numpy(v).to(dev) for k, v in lay["tables"].items()},
                      wp=torch.from_numpy(pack_conv_weights(w.cpu().numpy(), lay["cp"])).to(dev),
                      a=torch.empty(nb * lay["hp"] * lay["wp"] * lay["cp"], dtype=torch.float32, device=dev),
                      c=torch.empty(lay["M"] * lay["ncs"], dtype=torch.float32, device=dev))
            pr = (_lib.GGProblem * 1)()
            q, t = pr[0], st["tabs"]
            q.A, q.B, q.C, q.bias, q.R = st["a"].data_ptr(), st["wp"].data_ptr(), st["c"].data_ptr(), None, None
            q.rowA, q.colA, q.rowB, q.colB = t["rowA"].data_ptr(), t["colA"].data_ptr(), t["rowB"].data_ptr(), t["colB"].data_ptr()
            q.rowC, q.colC, q.rowR = t["rowC"].data_ptr(), t["colC"].data_ptr(), None
            q.M, q.N, q.K, q.tilesM, q.tilesN = lay["M"], lay["N"], lay["K"], lay["tiles_m"], lay["tiles_n"]
            q.splitK, q.chunksPerSplit, q.act, q.alpha, q.splitStride = 1, lay["K"] // 32, 0, 1.0, 0
            plan = C.c_void_p()
            check(lib.vsr_gemm_plan_create(pr, 1, lay["tile_cfg"], BMODE_NK, 3, C.byref(plan)))
            st["plan"] = plan
            self._gemm[key] = st
        lay = st["lay"]
        self._call(lib.vsr_det_launch_nchw_to_nhwc, _p(xin), nb, cin, h, wd, pt, pl, lay["hp"], lay["wp"], lay["cp"], _p(st["a"]), self._sa)
        self._call(lib.vsr_gemm_plan_run, st["plan"], self._sa)
        affine, act = self._fuse.get(i, (None, None))
        scale = shift = None
        if affine is not None and affine[0] == "bn":
            scale, shift = self.bn[affine[1]]
        elif affine is not None:
            shift = self.params[affine[2]].reshape(-1)
            if shift.numel() != lay["N"]:
                affine = act = None
                shift = None
            else:
                scale = self._ones(lay["N"])
        out = self._new(nb, lay["N"], ho, wo)
        self._call(lib.vsr_det_launch_nhwc_to_nchw, _p(st["c"]), nb, lay["N"], lay["P"], lay["ncs"], _p(scale), _p(shift), act[1] if act is not None else 0,
                                              _p(out), self._sa)
        return out, [j for j in (affine[1] if affine is not None else None, act[0] if act is not None else None) if j is not None]

    def _deconv_gemm(self, i, xin, w):
        """conv2d_transpose 2x2 / stride 2 (dense, Cin and Cout whole 32-float chunks) as ONE gather-GEMM: every input pixel is a row,
        the columns are (dy, dx, cout) -- out[2y+dy][2x+dx][co] = sum_ci x[y][x][ci] * W[ci][co][dy][dx] -- and the C tables scatter the
        four taps of a row to their places in the NHWC image [n][2h][2w][cout]; the bias add that follows rides on the GEMM's bias, the
        batch_norm / activation on the pass that writes the NCHW result.  (The direct kernel spent 5.3 ms on the 64 -> 64 deconv of a
        16-frame forward, profiles/r05_detector_kernel_stats.csv: one thread per output element looping over Cin.)
        Returns (output NCHW, op indices folded into it)."""
        nb, cin, h, wd = xin.shape
        cout = int(w.shape[1])
        key = ("deconv", i, tuple(xin.shape))
        bias, bn, act = self._dfuse.get(i, (None, None, None))
        st = self._gemm.get(key)
        if st is None:
            dev = self.device
            M, N, K = nb * h * wd, 4 * cout, cin
            bm, bn_, cfg = 128, 128, TILE_128x128
            tiles_m, tiles_n = -(-M // bm), -(-N // bn_)
            if nb * 4 * h * wd * cout >= 2 ** 31 or M * cin >= 2 ** 31:
                raise ValueError("transposed convolution too large for 32-bit offset tables")
            img, pix = np.divmod(np.arange(M, dtype=np.int64), h * wd)
            y, x = np.divmod(pix, wd)
            row_a = np.zeros(tiles_m * bm, np.int64)
            row_a[:M] = np.arange(M, dtype=np.int64) * cin
            row_c = np.zeros(tiles_m * bm, np.int64)
            row_c[:M] = ((img * 2 * h + 2 * y) * 2 * wd + 2 * x) * cout
            tap, cc = np.divmod(np.arange(tiles_n * bn_ // 32, dtype=np.int64), cout // 32)
            tap = np.minimum(tap, 3)                                    # (columns beyond N are never stored)
            col_c = ((tap // 2) * 2 * wd + tap % 2) * cout + cc * 32
            row_b = np.zeros(tiles_n * bn_, np.int64)
            row_b[:N] = np.arange(N, dtype=np.int64) * K
            tabs = dict(rowA=row_a, colA=np.arange(K // 32, dtype=np.int64) * 32, rowB=row_b, colB=np.arange(K // 32, dtype=np.int64) * 32,
                        rowC=row_c, colC=col_c)
            wn = np.asarray(w.cpu().numpy(), np.float32)                # [cin][cout][2][2] -> B[(dy, dx, co)][ci]
            wp = np.ascontiguousarray(wn.transpose(2, 3, 1, 0).reshape(N, K))
            bvec = np.zeros(tiles_n * bn_, np.float32)
            if bias is not None:
                bvec[:N] = np.tile(self.params[bias[1]].reshape(-1).cpu().numpy().astype(np.float32), 4)
            st = dict(tabs={k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in tabs.items()}, wp=torch.from_numpy(wp).to(dev),
                      bias=torch.from_numpy(bvec).to(dev), a=torch.empty(M * cin, dtype=torch.float32, device=dev),
                      c=torch.empty(nb * 4 * h * wd * cout, dtype=torch.float32, device=dev), M=M, N=N)
            pr = (_lib.GGProblem * 1)()
            q, t = pr[0], st["tabs"]
            q.A, q.B, q.C, q.bias, q.R = st["a"].data_ptr(), st["wp"].data_ptr(), st["c"].data_ptr(), st["bias"].data_ptr() if bias is not None else None, None
            q.rowA, q.colA, q.rowB, q.colB = t["rowA"].data_ptr(), t["colA"].data_ptr(), t["rowB"].data_ptr(), t["colB"].data_ptr()
            q.rowC, q.colC, q.rowR = t["rowC"].data_ptr(), t["colC"].data_ptr(), None
            q.M, q.N, q.K, q.tilesM, q.tilesN = M, N, K, tiles_m, tiles_n
            q.splitK, q.chunksPerSplit, q.act, q.alpha, q.splitStride = 1, K // 32, 0, 1.0, 0
            plan = C.c_void_p()
            check(lib.vsr_gemm_plan_create(pr, 1, cfg, BMODE_NK, 3, C.byref(plan)))
            st["plan"] = plan
            self._gemm[key] = st
        self._call(lib.vsr_det_launch_nchw_to_nhwc, _p(xin), nb, cin, h, wd, 0, 0, h, wd, cin, _p(st["a"]), self._sa)
        self._call(lib.vsr_gemm_plan_run, st["plan"], self._sa)
        scale = shift = None
        if bn is not None:
            scale, shift = self.bn[bn]
        out = self._new(nb, cout, 2 * h, 2 * wd)
        self._call(lib.vsr_det_launch_nhwc_to_nchw, _p(st["c"]), nb, cout, 4 * h * wd, cout, _p(scale), _p(shift), act[1] if act is not None else 0,
                   _p(out), self._sa)
        done = [j for j in (bias[0] if bias is not None else None, bn, act[0] if act is not None else None) if j is not None]
        return out, done

    def _ones(self, n):
        if n not in self._one:
            self._one[n] = torch.ones(n, dtype=torch.float32, device=self.device)
        return self._one[n]

    def _new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _binary(self, a, b, op):
        if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.numel() < b.numel():
            a, b = b, a                                            # add / multiply commute: keep the full tensor first
        n, c = a.shape[0], a.shape[1]
        hw = a.numel() // (n * c)
        if tuple(b.shape) == tuple(a.shape):
            mode = 0
        elif b.numel() == 1:
            mode = 3                                               # LearnableAffineBlock's scalar scale / bias
        elif b.numel() == c:
            mode = 1
        elif b.numel() == n * c:
            mode = 2
        else:
            raise NotImplementedError(f"broadcast {tuple(a.shape)} with {tuple(b.shape)}")
        out = self._new(*a.shape)
        b = self._c(b)
        if self._tape is not None:
            self._keep.append(b)
        self._call(lib.vsr_det_launch_binary, _p(a), _p(b), op, a.numel(), c, hw, mode, _p(out), self._sa)
        return out

    def _unary(self, x, kind, p0=0.0, p1=0.0):
        out = self._new(*x.shape)
        self._call(lib.vsr_det_launch_unary, _p(x), x.numel(), kind, C.c_float(p0), C.c_float(p1), _p(out), self._sa)
        return out

    def run(self, x):
        """x: fp32 [N,3,H,W] on the device -> probability map [N,1,H,W]"""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        val = dict(self.params)
        val[self.graph.input_id] = x
        folded = {}
        fl = 0.0
        self._sa.value = torch.cuda.current_stream().cuda_stream
        if self._tape is not None:
            self._keep.append(val)                              # a recorded pass keeps every intermediate alive
        with torch.cuda.device(self.device):
            for i, (kind, ins, outs, a) in enumerate(self.graph.ops):
                g = lambda j: val[ins[j]]
                if i in folded:                  # batch_norm / bias add / activation already applied by the conv's output pass
                    val[outs[0]] = folded[i]
                    continue
                if kind in ("conv2d", "depthwise_conv2d"):
                    xin, w = self._c(g(0)), g(1)
                    n, cin, h, wd = xin.shape
                    cout, _, kh, kw = w.shape
                    sh, sw = a["strides"]
                    pt, pl = a["paddings"][0], a["paddings"][1]
                    if a.get("padding_algorithm") == "SAME":
                        dil = list(a.get("dilations", [1, 1]))
                        (pt, ho), (pl, wo) = same_padding(h, kh, sh, dil[0]), same_padding(wd, kw, sw, dil[1])
                    else:
                        ho, wo = (h + 2 * pt - kh) // sh + 1, (wd + 2 * pl - kw) // sw + 1
                    dw = 1 if a["groups"] == cin and a["groups"] > 1 else 0
                    if not dw and a["groups"] != 1:
                        raise NotImplementedError("grouped conv")
                    fl += 2.0 * n * cout * ho * wo * kh * kw * (1 if dw else cin)
                    if (self.use_gemm and not dw and cin * kh * kw >= GEMM_MIN_K and list(a.get("dilations", [1, 1])) == [1, 1]):
                        out, done = self._conv_gemm(i, xin, w, (sh, sw), pt, pl, ho, wo)
                        folded.update({j: out for j in done})
                        val[outs[0]] = out
                        continue
                    out = self._new(n, cout, ho, wo)
                    self._call(lib.vsr_det_launch_conv2d, _p(xin), _p(w), None, n, cin, h, wd, cout, kh, kw, sh, sw, pt, pl, ho, wo, dw, 0, _p(out), self._sa)
                    val[outs[0]] = out
                elif kind == "conv2d_transpose":
                    xin, w = self._c(g(0)), g(1)
                    n, cin, h, wd = xin.shape
                    if tuple(w.shape[2:]) != (2, 2) or list(a["strides"]) != [2, 2] or list(a["paddings"]) != [0, 0]:
                        raise NotImplementedError("conv2d_transpose other than 2x2 / stride 2")
                    dw = 1 if a["groups"] == cin and a["groups"] > 1 else 0
                    cout = cin if dw else w.shape[1]
                    fl += 2.0 * n * h * wd * 4 * cout * (1 if dw else cin)
                    if self.use_gemm and self.deconv_gemm and not dw and a["groups"] == 1 and cin % 32 == 0 and cout % 32 == 0:
                        out, done = self._deconv_gemm(i, xin, w)
                        # a chain is folded only as a whole prefix: bias alone, bias + bn, bias + bn + act (deconv_fusions builds it so)
                        folded.update({j: out for j in done})
                        val[outs[0]] = out
                        continue
                    out = self._new(n, cout, 2 * h, 2 * wd)
                    self._call(lib.vsr_det_launch_deconv2x2, _p(xin), _p(w), n, cin, h, wd, cout, dw, _p(out), self._sa)
                    val[outs[0]] = out
                elif kind == "batch_norm_":
                    xin = self._c(g(0))
                    s, t = self.bn[i]
                    out = self._new(*xin.shape)
                    self._call(lib.vsr_det_launch_affine, _p(xin), _p(s), _p(t), xin.numel(), xin.shape[1], xin.shape[2] * xin.shape[3], _p(out), self._sa)
                    val[outs[0]] = out
                elif kind == "full_int_array":
                    val[outs[0]] = [int(v) for v in a["value"]]
                elif kind == "full":
                    val[outs[0]] = a["value"]
                elif kind == "reshape":
                    val[outs[0]] = self._c(g(0)).reshape(g(1))
                elif kind == "add":
                    val[outs[0]] = self._binary(g(0), g(1), 0)
                elif kind == "multiply":
                    val[outs[0]] = self._binary(g(0), g(1), 1)
                elif kind == "relu":
                    val[outs[0]] = self._unary(self._c(g(0)), 0)
                elif kind == "hardswish":
                    val[outs[0]] = self._unary(self._c(g(0)), 1)
                elif kind == "hardsigmoid":
                    val[outs[0]] = self._unary(self._c(g(0)), 2, a["slope"], a["offset"])
                elif kind == "sigmoid":
                    val[outs[0]] = self._unary(self._c(g(0)), 3)
                elif kind == "scale":
                    s = val[ins[1]] if len(ins) > 1 and ins[1] in val else a.get("scale", 1.0)
                    s, b = float(s), float(a.get("bias", 0.0))
                    val[outs[0]] = self._unary(self._c(g(0)), 4, s, b if a.get("bias_after_scale", True) else b * s)
                elif kind == "pool2d":
                    xin, ks = self._c(g(0)), g(1)
                    n, c, h, wd = xin.shape
                    if a["adaptive"]:
                        if list(ks) != [1, 1] or a["pooling_type"] != "avg":
                            raise NotImplementedError("adaptive pool other than global average")
                        out = self._new(n, c, 1, 1)
                        self._call(lib.vsr_det_launch_gap, _p(xin), n * c, h * wd, _p(out), self._sa)
                    else:
                        if a["pooling_type"] != "max":
                            raise NotImplementedError("average pool")
                        sh, sw = a["strides"]
                        pt, pl = a["paddings"][0], a["paddings"][1]
                        if a.get("padding_algorithm") == "SAME":
                            (pt, ho), (pl, wo) = same_padding(h, ks[0], sh), same_padding(wd, ks[1], sw)
                        elif a["ceil_mode"]:
                            ho, wo = -(-(h + 2 * pt - ks[0]) // sh) + 1, -(-(wd + 2 * pl - ks[1]) // sw) + 1
                        else:
                            ho, wo = (h + 2 * pt - ks[0]) // sh + 1, (wd + 2 * pl - ks[1]) // sw + 1
                        out = self._new(n, c, ho, wo)
                        self._call(lib.vsr_det_launch_maxpool, _p(xin), n * c, h, wd, ks[0], ks[1], sh, sw, pt, pl, ho, wo, _p(out), self._sa)
                    val[outs[0]] = out
                elif kind == "nearest_interp":
                    xin = self._c(g(0))
                    n, c, h, wd = xin.shape
                    s = int(a["scale"][0])
                    if a["scale"][0] != a["scale"][1] or s != a["scale"][0]:
                        raise NotImplementedError("non-integer nearest_interp scale")
                    out = self._new(n, c, h * s, wd * s)
                    self._call(lib.vsr_det_launch_nearest, _p(xin), n * c, h, wd, s, _p(out), self._sa)
                    val[outs[0]] = out
                elif kind == "combine":
                    val[outs[0]] = [val[j] for j in ins]
                elif kind == "concat":
                    parts, dim = g(0), int(g(1))
                    if dim == 1 and all(t.is_contiguous() for t in parts):      # channel concat: one strided block copy per part
                        nb = parts[0].shape[0]
                        out = self._new(nb, sum(t.shape[1] for t in parts), *parts[0].shape[2:])
                        pitch, at = out.numel() // nb * 4, 0
                        for t in parts:
                            w_ = t.numel() // nb * 4
                            self._call(lib.vsr_det_launch_copy, _p(t), w_, C.c_void_p(out.data_ptr() + at), pitch, w_, nb, self._sa)
                            at += w_
                        val[outs[0]] = out
                    else:
                        if self._tape is not None:
                            raise NotImplementedError("recorded replay of a concat that is not a channel concat")
                        val[outs[0]] = torch.cat(parts, dim=dim)            # pure data movement
                else:
                    raise NotImplementedError(f"detector op {kind}")
        self.flops[tuple(x.shape)] = fl
        return val[self.graph.output_id]


# ------------------------------------------------------------------------------------------------
# DBPostProcess (inference.yml PostProcess: thresh 0.3, box_thresh 0.6, max_candidates 1000, unclip_ratio 1.5) as PaddleX runs it
# behind TextDetection.predict (box_type "quad", score_mode "fast"): borders of the thresholded map (cv2.findContours, RETR_LIST:
# outer borders AND hole borders, last found first) -> minimum-area rectangle of every border -> mean probability over the
# rectangle rasterised the way cv2.fillPoly does it (integer-truncated corners, outline drawn, scan lines between) ->
# pyclipper's rounded polygon offset of the integer-truncated rectangle by area * ratio / perimeter -> minimum-area rectangle of
# the offset polygon -> rescale, round half to even, clip.  Round 3: the first version scored by pixel-centre containment, grew
# the rectangle analytically and ignored hole borders; against the restated reference (oracle/db_postprocess.py, tests/
# test_db_postprocess.py) its scores were off by up to 0.1 -- boxes near box_thresh came and went -- and corners by up to three
# source pixels.  This is the host statement; DeviceDBPostProcess below runs the same steps on the GPU.
# ------------------------------------------------------------------------------------------------
def _convex_hull(pts):
    pts = sorted(set(map(tuple, pts)))
    if len(pts) <= 2:
        return np.array(pts, dtype=np.float64)

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return np.array(lower[:-1] + upper[:-1], dtype=np.float64)


def min_area_rect(points):
    """minimum-area enclosing rectangle of integer points (cv2.minAreaRect's definition) -> (4 corners [4,2], width, height)"""
    hull = _convex_hull(points)
    if len(hull) == 1:
        return np.repeat(hull, 4, axis=0), 0.0, 0.0
    if len(hull) == 2:
        return np.array([hull[0], hull[0], hull[1], hull[1]]), float(np.hypot(*(hull[1] - hull[0]))), 0.0
    # rotating calipers over all hull edges at once: project the hull on every edge direction u and its normal v
    e = np.roll(hull, -1, axis=0) - hull
    nrm = np.hypot(e[:, 0], e[:, 1])
    u = e / nrm[:, None]
    v = np.stack([-u[:, 1], u[:, 0]], 1)
    pu = hull[:, 0:1] * u[None, :, 0] + hull[:, 1:2] * u[None, :, 1]   # [points, edges]
    pv = hull[:, 0:1] * v[None, :, 0] + hull[:, 1:2] * v[None, :, 1]
    umin, umax, vmin, vmax = pu.min(0), pu.max(0), pv.min(0), pv.max(0)
    w, h = umax - umin, vmax - vmin
    i = int(np.argmin(w * h))                                         # first minimum, as a strict '<' scan finds it
    c = [umin[i] * u[i] + vmin[i] * v[i], umax[i] * u[i] + vmin[i] * v[i], umax[i] * u[i] + vmax[i] * v[i], umin[i] * u[i] + vmax[i] * v[i]]
    return np.array(c), float(w[i]), float(h[i])


def _order_box(c):
    """get_mini_boxes: the corners as float32 (cv2.boxPoints), sorted by x; of the left pair the upper one first, of the right pair
    the upper one second -> top-left, top-right, bottom-right, bottom-left (ties exactly as the reference's comparisons break them)"""
    p = sorted(np.asarray(c, np.float32).tolist(), key=lambda q: q[0])
    a, d = (0, 1) if p[1][1] > p[0][1] else (1, 0)
    b, e = (2, 3) if p[3][1] > p[2][1] else (3, 2)
    return np.array([p[a], p[b], p[e], p[d]], dtype=np.float32)


def trace_borders(bitmap):
    """cv2.findContours(bitmap, RETR_LIST, ...): every border of the 8-connected foreground by border following (Suzuki & Abe),
    outer and hole borders alike, in cv2's order (the border found last by the raster scan comes first) -> list of int [n,2] (x, y).
    Straight runs keep their interior points (CHAIN_APPROX_SIMPLE would drop them; the rectangle fit only sees the hull).
    The walk itself is vsr_host_trace_borders (csrc/det_host.cpp): host code, like the reference's, a millisecond per map."""
    bm = np.ascontiguousarray(bitmap, dtype=np.uint8)
    H, W = bm.shape
    cap_p, cap_b = max(4096, 2 * int(bm.sum()) + 16), max(1024, int(bm.sum()) + 16)
    while True:
        pts = np.empty((cap_p, 2), np.int32)
        start = np.empty(cap_b + 1, np.int64)
        nb, npts = C.c_int32(0), C.c_int64(0)
        rc = lib.vsr_host_trace_borders(bm.ctypes.data_as(C.c_void_p), H, W, pts.ctypes.data_as(C.c_void_p), cap_p, start.ctypes.data_as(C.c_void_p),
                                        cap_b, C.byref(nb), C.byref(npts))
        if rc == -100:                                                  # a border can pass a pixel more than twice: grow and repeat
            cap_p, cap_b = max(cap_p, int(npts.value)), max(cap_b, int(nb.value))
            continue
        check(rc)
        break
    n = int(nb.value)
    start[n] = int(npts.value)
    return [pts[start[k]:start[k + 1]] for k in range(n - 1, -1, -1)]


def _line_mask(xx, yy, x0, y0, x1, y1):
    """pixels of the 8-connected line cv2 draws from (x0, y0) to (x1, y1) (LineIterator, left to right), as a closed form of its
    error recurrence: along the major axis the minor coordinate has advanced max(0, (2 * minor * k + major - 1) // (2 * major)) after k steps"""
    if x1 < x0:
        x0, y0, x1, y1 = x1, y1, x0, y0
    dx, dy = x1 - x0, abs(y1 - y0)
    sy = 1 if y1 >= y0 else -1
    if dy > dx:
        k = (yy - y0) * sy
        return (k >= 0) & (k <= dy) & (xx == x0 + np.maximum(0, (2 * dx * k + dy - 1) // (2 * dy)))
    k = xx - x0
    adv = np.maximum(0, (2 * dy * k + dx - 1) // (2 * dx)) if dx else 0
    return (k >= 0) & (k <= dx) & (yy == y0 + sy * adv)


def fill_poly_mask(quad, h, w):
    """cv2.fillPoly(zeros((h, w)), [quad], 1) for a convex quadrilateral with integer vertices: the outline as 8-connected lines plus,
    for every scan line y in [top, bottom) of the two edges that span it, the pixels floor(x_left) .. floor(x_right) with the edge
    positions in 16.16 fixed point advancing by the truncated per-row slope"""
    yy, xx = np.mgrid[0:h, 0:w]
    m = np.zeros((h, w), bool)
    rows = np.arange(h)
    lo = np.full(h, np.iinfo(np.int64).max, np.int64)
    hi = np.full(h, np.iinfo(np.int64).min, np.int64)
    for a in range(4):
        x0, y0, x1, y1 = int(quad[a - 1][0]), int(quad[a - 1][1]), int(quad[a][0]), int(quad[a][1])
        m |= _line_mask(xx, yy, x0, y0, x1, y1)
        if y0 == y1:
            continue
        num, den = (x1 - x0) << 16, y1 - y0
        slope = (abs(num) // abs(den)) * (1 if (num >= 0) == (den > 0) else -1)          # C integer division: toward zero
        ya, yb, xa = (y0, y1, x0) if y0 < y1 else (y1, y0, x1)
        live = (rows >= ya) & (rows < yb)
        xf = (xa << 16) + (rows - ya).astype(np.int64) * slope
        lo = np.where(live, np.minimum(lo, xf), lo)
        hi = np.where(live, np.maximum(hi, xf), hi)
    span = (hi >= lo)
    m |= span[:, None] & (xx >= (lo >> 16)[:, None]) & (xx <= (hi >> 16)[:, None])
    return m


def _box_score(prob_of, box, H, W):
    """box_score_fast: mean probability over cv2.fillPoly's raster of the (integer-truncated) box inside its bounding rows / columns"""
    xa, xb = int(np.clip(np.floor(box[:, 0].min()), 0, W - 1)), int(np.clip(np.ceil(box[:, 0].max()), 0, W - 1))
    ya, yb = int(np.clip(np.floor(box[:, 1].min()), 0, H - 1)), int(np.clip(np.ceil(box[:, 1].max()), 0, H - 1))
    q = (box - np.array([xa, ya], np.float32)).astype(np.int32)                        # float32 subtraction, truncation toward zero
    mask = fill_poly_mask(q, yb - ya + 1, xb - xa + 1)
    crop = prob_of(ya, yb + 1, xa, xb + 1)
    return float(crop[mask].astype(np.float64).mean()) if mask.any() else 0.0


def _away(v):
    return int(v - 0.5) if v < 0 else int(v + 0.5)


def offset_polygon_round(path, delta):
    """pyclipper.PyclipperOffset().AddPath(path, JT_ROUND, ET_CLOSEDPOLYGON); Execute(delta) for a convex polygon, delta > 0
    (ClipperLib 6.4.2: integer-truncated input, unit normals, a corner replaced by an arc of round(steps_per_radian * angle) chords for
    an arc tolerance of 0.25, every vertex rounded half away from zero) -> list of integer (x, y)"""
    pts = []
    for x, y in path:
        q = (int(x), int(y))
        if not pts or q != pts[-1]:
            pts.append(q)
    if len(pts) > 1 and pts[0] == pts[-1]:
        pts.pop()
    n = len(pts)
    if n < 3:
        return []
    twice = sum((pts[i - 1][0] + pts[i][0]) * (pts[i - 1][1] - pts[i][1]) for i in range(n))
    if twice > 0:                                                     # Clipper's Area() = -twice / 2 < 0: reversed, normals then point outwards
        pts.reverse()
    tol = min(0.25, delta * 0.25)
    steps = min(np.pi / np.arccos(1 - tol / delta), delta * np.pi)
    rot_s, rot_c, per_rad = np.sin(2 * np.pi / steps), np.cos(2 * np.pi / steps), steps / (2 * np.pi)
    nrm = []
    for i in range(n):
        ex, ey = pts[(i + 1) % n][0] - pts[i][0], pts[(i + 1) % n][1] - pts[i][1]
        inv = 1.0 / np.sqrt(ex * ex + ey * ey)
        nrm.append((ey * inv, -ex * inv))
    res = []
    put = lambda px, py, nx, ny: res.append((_away(px + nx * delta), _away(py + ny * delta)))
    for i in range(n):
        (ax, ay), (bx, by) = nrm[i - 1], nrm[i]
        sin_t, cos_t = ax * by - bx * ay, ax * bx + ay * by
        if abs(sin_t * delta) < 1.0 and cos_t > 0:                    # (almost) straight on: one vertex
            put(*pts[i], ax, ay)
            continue
        if abs(sin_t * delta) >= 1.0:
            sin_t = float(np.clip(sin_t, -1.0, 1.0))
        if sin_t * delta < 0:                                         # concave corner
            put(*pts[i], ax, ay)
            res.append(pts[i])
            put(*pts[i], bx, by)
            continue
        ang = np.arctan2(sin_t, cos_t)
        cx, cy = ax, ay
        for _ in range(max(_away(per_rad * abs(ang)), 1)):
            put(*pts[i], cx, cy)
            cx, cy = cx * rot_c - rot_s * cy, cx * rot_s + cy * rot_c
        put(*pts[i], bx, by)
    return res


def box_from_border(border, prob_of, H, W, src_h, src_w, box_thresh, unclip_ratio, min_size):
    """one iteration of DBPostProcess.boxes_from_bitmap: border points int [n,2] (x, y); prob_of(ya, yb, xa, xb) -> probability crop.
    -> (box int16 [4,2] in source pixels, score) or None"""
    # a rectangle fit costs a hull; specks are the bulk of a noisy map: the fitted rectangle is never larger in area than the
    # axis-aligned bounding box, so its short side is at most sqrt(box area)
    ext = border.max(0) - border.min(0)
    if float(ext[0]) * float(ext[1]) < float(min_size) * float(min_size):
        return None
    corners, w, h = min_area_rect(border)
    if min(w, h) < min_size:
        return None
    box = _order_box(corners)
    score = _box_score(prob_of, box, H, W)
    if box_thresh > score:
        return None
    x, y = box[:, 0].astype(np.float64), box[:, 1].astype(np.float64)
    area = abs(float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))) * 0.5          # cv2.contourArea
    seg = box - np.roll(box, 1, axis=0)                                               # cv2.arcLength: float32 segment lengths
    length = float(np.sqrt(seg[:, 0] * seg[:, 0] + seg[:, 1] * seg[:, 1]).astype(np.float64).sum())
    if area == 0.0 or length == 0.0:
        return None
    grown = offset_polygon_round(box.tolist(), area * unclip_ratio / length)
    if not grown:
        return None
    corners, w, h = min_area_rect(np.array(grown))
    if min(w, h) < min_size + 2:
        return None
    big = _order_box(corners).astype(np.float64)
    big[:, 0] = np.clip(np.round(big[:, 0] * (src_w / W)), 0, src_w)                  # np.round: half to even, as Python's round()
    big[:, 1] = np.clip(np.round(big[:, 1] * (src_h / H)), 0, src_h)
    return big.astype(np.int16), score


def db_postprocess(prob, src_h, src_w, thresh=0.3, box_thresh=0.6, max_candidates=1000, unclip_ratio=1.5, min_size=3):
    """prob [H,W] (host array) -> (boxes int16 [n,4,2] in source-image pixels, scores [n]); everything on the host"""
    prob = np.asarray(prob, np.float32)
    H, W = prob.shape
    whole = lambda ya, yb, xa, xb: prob[ya:yb, xa:xb]
    boxes, scores = [], []
    for border in trace_borders(prob > thresh)[:max_candidates]:
        r = box_from_border(border, whole, H, W, src_h, src_w, box_thresh, unclip_ratio, min_size)
        if r is not None:
            boxes.append(r[0])
            scores.append(r[1])
    return (np.stack(boxes) if boxes else np.zeros((0, 4, 2), np.int16)), scores


class DeviceDBPostProcess:
    """DBPostProcess on the GPU, on the probability map where the forward left it (vsr_det_launch_db_boxes): threshold, 8-connected
    labelling (union-find), per-component bounding box, the hole count (Euler number), then per component -- one workgroup -- the
    steps of box_from_border: hull of its per-row extreme pixels, minimum-area rectangle, cv2.fillPoly-rule score, ClipperLib-rule
    offset, second rectangle fit, rescale.  The host receives one buffer: a header and a 16-int record per component, i.e. one
    synchronisation per frame.  Held to oracle/db_postprocess.py in tests/test_gpu_ocr_det.py (boxes equal, scores to 1e-6).
    The device handles what a subtitle frame looks like; the map goes to the host statement (db_postprocess on the downloaded
    map) when it has a HOLE (cv2.findContours lists hole borders too, and a component's outer border is then not the only contour
    it contributes), more than `cap` components (a noise map), a component taller than 256 rows, or an offset polygon beyond
    the kernel's point buffer."""

    REC, HDR = 16, 4

    def __init__(self, device, cap=256):
        self.device, self.cap = device, cap
        self._work = {}
        self.host_fallbacks = 0

    def _buffers(self, H, W, slot=0):
        key = (H, W, slot)
        if key not in self._work:
            dev, i32 = self.device, torch.int32
            self._work[key] = (torch.empty(H * W, dtype=i32, device=dev), torch.empty(H * W * 5, dtype=i32, device=dev),
                               torch.empty(self.cap * 6, dtype=i32, device=dev), torch.zeros(4, dtype=i32, device=dev),
                               torch.empty(self.cap * H * 2, dtype=i32, device=dev), torch.zeros(self.HDR + self.cap * self.REC, dtype=i32, device=dev))
        return self._work[key]

    def _launch(self, prob_dev, src_h, src_w, thresh, box_thresh, unclip_ratio, min_size, slot=0):
        """the post-process kernels of one map on the current stream, in the work buffers of `slot`; -> (header + records, components)"""
        H, W = prob_dev.shape
        labels, stats, comps, count, ext, out = self._buffers(H, W, slot)
        check(lib.vsr_det_launch_db_boxes(_p(prob_dev), H, W, C.c_float(thresh), src_h, src_w, C.c_float(box_thresh), C.c_float(unclip_ratio),
                                          min_size, _p(labels), _p(stats), _p(comps), _p(count), _p(ext), _p(out), self.cap, _stream()))
        return out, comps

    def _decode(self, host, prob_dev, src_h, src_w, thresh, box_thresh, max_candidates, unclip_ratio, min_size):
        """boxes and scores from the downloaded header + records + components of one map"""
        n, holes = int(host[0]), int(host[1])
        if n == 0:
            return np.zeros((0, 4, 2), np.int16), []
        rec = host[self.HDR:self.HDR + self.cap * self.REC].reshape(self.cap, self.REC)[:min(n, self.cap)]
        if holes != 0 or n > self.cap or (rec[:, 0] < 0).any():
            self.host_fallbacks += 1
            return db_postprocess(prob_dev.cpu().numpy(), src_h, src_w, thresh, box_thresh, max_candidates, unclip_ratio, min_size)
        first = host[self.HDR + self.cap * self.REC:].reshape(self.cap, 6)[:n, 0]
        order = np.argsort(first)[::-1][:max_candidates]       # cv2.findContours lists the border found LAST by the raster scan first
        rec = rec[order]
        rec = rec[rec[:, 0] == 1]
        boxes = rec[:, 1:9].reshape(-1, 4, 2).astype(np.int16)
        scores = [float(v) for v in np.ascontiguousarray(rec[:, 9]).view(np.float32)]
        return boxes, scores

    def __call__(self, prob_dev, src_h, src_w, thresh=0.3, box_thresh=0.6, max_candidates=1000, unclip_ratio=1.5, min_size=3):
        assert prob_dev.is_cuda and prob_dev.dtype == torch.float32 and prob_dev.dim() == 2
        prob_dev = prob_dev.contiguous()
        with torch.cuda.device(self.device):
            out, comps = self._launch(prob_dev, src_h, src_w, thresh, box_thresh, unclip_ratio, min_size)
            host = torch.cat([out, comps]).cpu().numpy()                             # the one synchronisation of the post-process
        return self._decode(host, prob_dev, src_h, src_w, thresh, box_thresh, max_candidates, unclip_ratio, min_size)

    def batch(self, probs_dev, src_h, src_w, thresh=0.3, box_thresh=0.6, max_candidates=1000, unclip_ratio=1.5, min_size=3):
        """[self(p, ...) for p in probs_dev] for the maps [n, H, W] of one forward with ONE synchronisation: every map is post-processed in
        work buffers of its own (slot = index in the batch), the n record buffers come down in one copy (a per-frame read-back leaves the
        GPU idle for a host round trip between the maps: 8 of them per forward)"""
        assert probs_dev.is_cuda and probs_dev.dtype == torch.float32 and probs_dev.dim() == 3
        n = probs_dev.shape[0]
        if n == 0:
            return []
        with torch.cuda.device(self.device):
            maps = [probs_dev[b].contiguous() for b in range(n)]
            parts = []
            for b in range(n):
                parts.extend(self._launch(maps[b], src_h, src_w, thresh, box_thresh, unclip_ratio, min_size, slot=b))
            host = torch.cat(parts).cpu().numpy().reshape(n, -1)
        return [self._decode(host[b], maps[b], src_h, src_w, thresh, box_thresh, max_candidates, unclip_ratio, min_size) for b in range(n)]


def det_resize_shape(H, W, limit_side_len=960, limit_type="max", max_side_limit=4000):
    """Net input size (rh, rw) of DetResizeForTest for an HxW image.

    inference.yml:31-32 only says `DetResizeForTest: resize_long: 960`.  paddleocr 3.4's TextDetection (PaddleX
    text_detection/predictor.py `build_resize`) maps that key, for the PP-OCRv5 det models, to
    `limit_side_len = resize_long`, `limit_type = "max"` and runs processors.DetResizeForTest.resize_image_type0:
    the image is only SHRUNK when its longer side exceeds the limit (ratio 1 otherwise), sides are truncated with int()
    and then rounded to multiples of 32 (minimum 32).  [external: PaddleX is not in the reference mount; restated from its
    published source, unverifiable here -- parity unpinned.]  `limit_type="long"` gives round 1's behaviour (longer side
    always scaled to the limit)."""
    if limit_type == "max":
        ratio = float(limit_side_len) / max(H, W) if max(H, W) > limit_side_len else 1.0
    elif limit_type == "min":
        ratio = float(limit_side_len) / min(H, W) if min(H, W) < limit_side_len else 1.0
    elif limit_type == "long":
        ratio = float(limit_side_len) / max(H, W)
    else:
        raise ValueError(f"limit_type {limit_type!r}")
    rh, rw = int(H * ratio), int(W * ratio)
    if max(rh, rw) > max_side_limit:
        r2 = float(max_side_limit) / max(rh, rw)
        rh, rw = int(rh * r2), int(rw * r2)
    return max(int(round(rh / 32) * 32), 32), max(int(round(rw / 32) * 32), 32)


def same_padding(size, k, s, d=1):
    """Paddle's padding_algorithm == "SAME" along one axis: (pad_before, out_size); the odd pixel goes after."""
    out = -(-size // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - size, 0)
    return total // 2, out


class TextDetection:
    def __init__(self, model, weights, device=0, resize_long=960, limit_type="max"):
        """model: directory holding inference.json (as backend/models/V5/ch_det), a path to it, or a loaded / condensed graph dict"""
        if isinstance(model, (str, os.PathLike)) and os.path.isdir(model):
            model = os.path.join(model, "inference.json")
        self.graph = model if hasattr(model, "ops") else load_graph(model)
        self._weights = weights
        self.runner = PaddleGraphRunner(self.graph, weights, device)
        self.resize_long = resize_long
        self.limit_type = limit_type
        # VSR_DET_GRAPH=1 replays the forward from a captured HIP graph.  OFF by default: the recorded launch list (run_taped) is as fast
        # (the forward is not launch-bound at the batch sizes used) and needs nothing from the driver.  The memory access fault of round 1
        # was traced in round 6 to the captured hipMemsetAsync nodes of the resident GEMM plans (profiles/r06_det_graph_triage.log):
        # the plans now zero their tile counters with a kernel and the replay is clean.
        self.use_graph = os.environ.get("VSR_DET_GRAPH", "0") == "1"
        # VSR_DET_TAPE=0 walks the program op by op on every frame (the recorded launch list is the default, see run_taped)
        self.use_tape = os.environ.get("VSR_DET_TAPE", "1") != "0"
        self.device = self.runner.device
        self._tables = {}

    def gflop_per_frame(self, rh, rw):
        """algorithmic GFLOP of one frame's forward at net input rh x rw, from the last walk of the program at that size (None before one)"""
        for shape, fl in self.runner.flops.items():
            if tuple(shape[2:]) == (rh, rw):
                return fl / shape[0] / 1e9
        return None

    def clone(self):
        """a second detector on the same device from the same program and weights: its own runner (recorded launch lists,
        intermediates) and post-process buffers -- one per lane of SubtitleDetect's resident pass (VSR_DET_LANES, tools/batch_lanes.py)"""
        return type(self)(self.graph, self._weights, device=self.device.index if hasattr(self.device, "index") else self.device,
                          resize_long=self.resize_long, limit_type=self.limit_type)

    def _resize(self, img_dev, H, W, rh, rw):
        """cv2.resize(img, (rw, rh)) INTER_LINEAR on uint8 -- the fixed-point kernel of the inpainting path"""
        key = (H, W, rh, rw)
        if key not in self._tables:
            tabs = []
            for ssize, dsize, clamp in ((W, rw, 1), (H, rh, 0)):
                ofs, ic, fc = np.zeros(dsize, np.int32), np.zeros(2 * dsize, np.int16), np.zeros(2 * dsize, np.float32)
                check(lib.vsr_cv2_linear_tables(ssize, dsize, clamp, ofs.ctypes.data_as(C.c_void_p), ic.ctypes.data_as(C.c_void_p), fc.ctypes.data_as(C.c_void_p)))
                tabs += [torch.from_numpy(ofs).to(self.device), torch.from_numpy(ic).to(self.device)]
            self._tables[key] = tabs
        xofs, ialpha, yofs, ibeta = self._tables[key]
        out = torch.empty((rh, rw, 3), dtype=torch.uint8, device=self.device)
        check(lib.vsr_launch_resize_u8(_p(img_dev), H * W * 3, W * 3, W, H, _p(out), rw, rh, 1, 3, None, _p(xofs), _p(ialpha), _p(yofs), _p(ibeta), _stream()))
        return out

    def probability_map(self, img):
        """img: HxWx3 uint8 BGR (numpy) -> (probability map [rh, rw] torch tensor on the device, rh, rw)"""
        H, W = img.shape[:2]
        rh, rw = det_resize_shape(H, W, self.resize_long, self.limit_type)
        with torch.cuda.device(self.device):
            d = torch.from_numpy(np.ascontiguousarray(img)).to(self.device)
            small = self._resize(d, H, W, rh, rw)
            x = torch.empty((1, 3, rh, rw), dtype=torch.float32, device=self.device)
            check(lib.vsr_det_launch_normalize(_p(small), rh, rw, _p(x), _stream()))
            prob = self.runner.run_graphed(x) if self.use_graph else (self.runner.run_taped(x) if self.use_tape else self.runner.run(x))
        return prob[0, 0], rh, rw

    def _post(self, prob_dev, src_h, src_w):
        if getattr(self, "_db", None) is None:
            self._db = DeviceDBPostProcess(self.device)
        if os.environ.get("VSR_DET_POST", "device") == "host":
            return db_postprocess(prob_dev.cpu().numpy(), src_h, src_w)
        return self._db(prob_dev, src_h, src_w)

    def _post_batch(self, probs_dev, src_h, src_w):
        """[_post(p) for p in probs_dev] with one read-back for the whole forward (DeviceDBPostProcess.batch; VSR_DET_POST_BATCH=0: per map)"""
        if getattr(self, "_db", None) is None:
            self._db = DeviceDBPostProcess(self.device)
        if os.environ.get("VSR_DET_POST", "device") == "host" or os.environ.get("VSR_DET_POST_BATCH", "1") == "0":
            return [self._post(probs_dev[b], src_h, src_w) for b in range(probs_dev.shape[0])]
        return self._db.batch(probs_dev, src_h, src_w)

    def predict(self, img):
        prob, _, _ = self.probability_map(img)
        boxes, scores = self._post(prob, img.shape[0], img.shape[1])
        return [{"dt_polys": boxes, "dt_scores": scores}]

    # how many sampled frames SubtitleDetect.find_subtitle_frame_no hands over at once (VSR_DET_BATCH).  One frame of the
    # server program is 32 640 GEMM rows at its finest stage and 510 at its coarsest: 256 CUs are filled by a batch, not by a
    # frame (16 -> 6 ms per frame measured, profiles/r02_detector_bench.log)
    batch_size = int(os.environ.get("VSR_DET_BATCH", "8"))

    def probability_maps(self, imgs):
        """imgs: n HxWx3 uint8 BGR frames of one size -> probability maps [n, rh, rw] on the device (one forward)"""
        n = len(imgs)
        H, W = imgs[0].shape[:2]
        rh, rw = det_resize_shape(H, W, self.resize_long, self.limit_type)
        with torch.cuda.device(self.device):
            d = torch.from_numpy(np.ascontiguousarray(np.stack(imgs))).to(self.device)
            prob = self._forward_frames(d, n, H, W, rh, rw)
        return prob[:n, 0]

    def _forward_frames(self, d, n, H, W, rh, rw):
        """resize + normalise + forward of n device frames.  A partial batch (the tail of a video) is padded to batch_size with copies of
        its last frame when launch lists are recorded: every image of a batch comes out exactly as it does alone, so the padding
        changes nothing, and the runner keeps ONE recorded list (and one set of intermediates) per frame size instead of one per
        tail length"""
        nb = self.batch_size if (self.use_tape and 1 < n < self.batch_size) else n
        x = torch.empty((nb, 3, rh, rw), dtype=torch.float32, device=self.device)
        for b in range(n):
            small = self._resize(d[b], H, W, rh, rw)
            check(lib.vsr_det_launch_normalize(_p(small), rh, rw, _p(x[b]), _stream()))
        if nb > n:
            x[n:] = x[n - 1]
        return self.runner.run_taped(x) if self.use_tape else self.runner.run(x)

    def probability_maps_device(self, frames_dev):
        """probability_maps for frames that already live in HBM (uint8 [n,H,W,3] BGR device tensor, tools/resident.py): no upload"""
        n, H, W, _ = frames_dev.shape
        rh, rw = det_resize_shape(H, W, self.resize_long, self.limit_type)
        with torch.cuda.device(self.device):
            prob = self._forward_frames(frames_dev.contiguous(), n, H, W, rh, rw)
        return prob[:n, 0]

    def predict_batch_device(self, frames_dev):
        """predict_batch on device frames: the same dicts"""
        if frames_dev.shape[0] == 0:
            return []
        prob = self.probability_maps_device(frames_dev)
        H, W = int(frames_dev.shape[1]), int(frames_dev.shape[2])
        return [{"dt_polys": boxes, "dt_scores": scores} for boxes, scores in self._post_batch(prob, H, W)]

    def predict_batch(self, imgs):
        """[predict(img)[0] for img in imgs] with one forward for all frames (independent per frame: same results)"""
        if len(imgs) == 0:
            return []
        prob = self.probability_maps(imgs)
        return [{"dt_polys": boxes, "dt_scores": scores} for boxes, scores in self._post_batch(prob, imgs[0].shape[0], imgs[0].shape[1])]


def from_env(device=0):
    """TextDetection from VSR_DET_MODEL_DIR (a directory like backend/models/V5/ch_det holding inference.json) and its weights:
    VSR_DET_WEIGHTS (an .npz of {parameter name: array}, or a .pdiparams file) or, when unset, inference.pdiparams inside the
    model directory (what paddleocr reads, subtitle_detect.py:41-54).  None when the directory or the weights are missing --
    callers then need an injected detector."""
    model_dir, wpath = os.environ.get("VSR_DET_MODEL_DIR"), os.environ.get("VSR_DET_WEIGHTS")
    if not model_dir:
        return None
    if not wpath:
        wpath = os.path.join(model_dir, "inference.pdiparams")
    if not os.path.isfile(wpath):
        return None
    return TextDetection(model_dir, load_weights(wpath, model_dir), device=device)


def load_weights(wpath, model_dir):
    if wpath.endswith(".npz"):
        with np.load(wpath) as z:
            return {k: z[k] for k in z.files}
    graph = load_graph(os.path.join(model_dir, "inference.json")) if os.path.isdir(model_dir) else load_graph(model_dir)
    return read_pdiparams(wpath, graph)
