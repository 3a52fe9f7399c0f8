"""NHWC-resident plan of the text detector's forward pass (reference call site
backend/tools/subtitle_detect.py:41-82, the program is backend/models/V5/{ch_det,ch_det_fast}/inference.json).

ocr_det.PaddleGraphRunner.run walks the program operator by operator on NCHW tensors; every dense conv on the gather-GEMM then
pays two layout passes (NCHW -> padded NHWC -> NCHW), every concat a block copy per part, every residual add and nearest_interp a
pass of its own: 30 % of the forward's GPU time (profiles/r06_detector_kernel_stats.csv).  This module COMPILES the program, once per
input shape, into a list of launches over buffers that stay NHWC between the convs:

  * a value produced by a GEMM conv / depthwise conv / transposed conv / nearest_interp / residual add lives in a zero-haloed NHWC
    buffer [n][H + 2 ph][W + 2 pw][Cs] whose halo is the largest padding any conv that reads it asks for: the consumer's A tables
    address it directly (no layout pass), the halo and the channels a slice is padded with are zero and nothing ever writes them;
  * batch_norm_ (scale into the weights, shift into the bias), the bias add and the ReLU that follow a conv ride on the GEMM epilogue;
    a same-shape add whose operand is a conv result becomes that GEMM's residual (C = act(..) + R), and when the other operand is a
    nearest_interp the residual rows address the low-resolution source (rowR): the FPN's upsample + add costs no launch;
  * a channel concat is its producers writing their slices of one buffer (slices padded to whole 32-float K chunks; the
    aggregation conv's weights get zero columns there);
  * the two one-channel convs of the DB head (1x1 conv, 2x2 transposed conv) are dot products per pixel with bias + sigmoid fused;
  * whatever has no NHWC form (the 3-channel stem, pooling, squeeze-excite gates, hardsigmoid ...) runs on the NCHW kernels as
    before, between a from_view / to_view pair.

The compiler is host code over numpy (no device): `compile_plan` returns buffers, constants and steps; ocr_det.PaddleGraphRunner
uploads and launches them, tests/_det_replay.py executes the same steps on the CPU against the program interpreter
(oracle/ppocr_det.py) -- the plan is checked without a GPU.
"""
import numpy as np

TILE_128x128, TILE_128x64 = 0, 3                 # include/vsr_hip.h VSR_TILE_*
GEMM_ACT_NONE, GEMM_ACT_RELU = 0, 2              # VSR_ACT_*
ACT_CODES = {None: 0, "relu": 1, "hardswish": 2, "sigmoid": 3}       # det_kernels.hip det_act / det_act4
GEMM_MIN_K = 64
TILE_256x32 = 1


def _opts():
    """tuning switches of the plan (environment, read at compile time; defaults = what profiles/r06c_det_plan_variants.log measured best):
    VSR_DET_N32_TILE=0: problems of 32 output columns on the 128 x 64 tile instead of the 256 x 32 one (a full N tile: float4 epilogue, no
        idle half of the MFMA columns)
    VSR_DET_THIN=0: every GEMM on the persistent LDS-DMA kernel (variant 3); default: short or few-tile problems (thin_variant below) on
        variant 1 -- one workgroup per tile, no tile queue: the persistent kernel's per-tile fixed cost is what the 48- / 96- / 32-channel
        blocks of PP-HGNetV2 were paying
    VSR_DET_GROUP=0: adjacent independent GEMMs are NOT merged into one launch
    VSR_DET_IM2COL=0: a one-channel part of a concat is NOT turned into an im2col chunk for the conv that reads it"""
    import os
    on = lambda k: os.environ.get(k, "1") != "0"
    return dict(n32=on("VSR_DET_N32_TILE"), thin=on("VSR_DET_THIN"), group=on("VSR_DET_GROUP"), im2col=on("VSR_DET_IM2COL"))


def thin_variant(M, N, K, tiles):
    """kernel variant of a GEMM problem: 1 (one workgroup per tile) for K <= 256, for N <= 96 with K <= 1152 and for launches of at most
    512 tiles with K <= 2400; 3 (persistent, LDS-DMA, pipelined tiles) otherwise -- the long-K convs of the neck and the head run at
    115-132 TF there (measured per conv at 8 frames of 960 x 544, profiles/r06c_det_plan_variants.log)"""
    if K <= 256 or (N <= 96 and K <= 1152) or (tiles <= 512 and K <= 2400):
        return 1
    return 3


def r32(c):
    return -(-int(c) // 32) * 32


def same_padding(size, k, s, d=1):
    out = -(-size // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - size, 0)
    return total // 2, out


class View:
    """channels [c0, c0 + physical extent) of the interior of the NHWC buffer `buf` = [n][H + 2 ph][W + 2 pw][Cs]; cmap lists where the
    logical channels sit: (logical start, count, physical start relative to c0) -- one entry unless the value is a concat"""

    def __init__(self, buf, n, H, W, C, ph, pw, Cs, c0=0, cmap=None):
        self.buf, self.n, self.H, self.W, self.C, self.ph, self.pw, self.Cs, self.c0 = buf, n, H, W, C, ph, pw, Cs, c0
        self.cmap = cmap or [(0, C, 0)]

    Hp = property(lambda s: s.H + 2 * s.ph)
    Wp = property(lambda s: s.W + 2 * s.pw)
    img_stride = property(lambda s: s.Hp * s.Wp * s.Cs)
    row_stride = property(lambda s: s.Wp * s.Cs)
    size = property(lambda s: s.n * s.Hp * s.Wp * s.Cs)
    cphys = property(lambda s: r32(s.cmap[-1][2] + s.cmap[-1][1]))
    gapped = property(lambda s: len(s.cmap) > 1)
    im2col = None        # {index into cmap: (kh, kw, pt, pl)}: that part's slice holds the kh x kw neighbourhoods of its channels (one chunk)

    def pix(self, img, y, x):
        """float offset (from the start of the buffer) of channel c0 of pixel (y, x) of image img; y, x may reach into the halo"""
        return ((img * self.Hp + y + self.ph) * self.Wp + x + self.pw) * self.Cs + self.c0

    origin = property(lambda s: s.pix(0, 0, 0))


class Val:
    """a value of the program while it is compiled: shape + where it lives (NCHW buffer and / or NHWC view), or a host constant"""

    def __init__(self, shape=None, nchw=None, view=None, const=None, vid=None):
        self.shape, self.nchw, self.view, self.const, self.vid = (tuple(shape) if shape is not None else None), nchw, view, const, vid
        self.deferred = None         # a GEMM conv waiting for the add that reads it (its residual)
        self.lazy_up = None          # (source Val, scale): a nearest_interp that only feeds such an add

    runtime = property(lambda s: s.const is None and s.shape is not None)


class Plan:
    def __init__(self):
        self.buffers = {}            # name -> [floats, zero-initialised?]
        self.consts = {}             # name -> numpy array (float32 weights / int32 tables)
        self.steps = []              # (kind, {..})
        self.input = None            # NCHW buffer the input is copied into
        self.output = None           # (buffer, shape)
        self.flops = 0.0
        self.stats = {}


def infer_shapes(graph, xshape):
    """tensor shapes / host constants of every value for one input shape"""
    shape, const = {graph.input_id: tuple(xshape)}, {}
    for v, (_, s) in graph.params.items():
        shape[v] = tuple(s)
    for kind, ins, outs, a in graph.ops:
        s = None
        if kind in ("conv2d", "depthwise_conv2d"):
            n, c, h, w = shape[ins[0]]
            co, _, kh, kw = shape[ins[1]]
            sh, sw = a["strides"]
            pt, pl = a["paddings"][0], a["paddings"][1]
            dil = list(a.get("dilations", [1, 1]))
            if a.get("padding_algorithm") == "SAME":
                ho, wo = same_padding(h, kh, sh, dil[0])[1], same_padding(w, kw, sw, dil[1])[1]
            else:
                ho, wo = (h + 2 * pt - kh) // sh + 1, (w + 2 * pl - kw) // sw + 1
            s = (n, co, ho, wo)
        elif kind == "conv2d_transpose":
            n, c, h, w = shape[ins[0]]
            dw = a["groups"] == c and a["groups"] > 1
            s = (n, c if dw else shape[ins[1]][1], 2 * h, 2 * w)
        elif kind == "full_int_array":
            const[outs[0]] = [int(v) for v in a["value"]]
            continue
        elif kind == "full":
            const[outs[0]] = a["value"]
            continue
        elif kind == "combine":
            const[outs[0]] = list(ins)
            continue
        elif kind == "reshape":
            tgt = list(const[ins[1]])
            numel = int(np.prod(shape[ins[0]]))
            if -1 in tgt:
                k = tgt.index(-1)
                tgt[k] = numel // int(-np.prod(tgt))
            s = tuple(tgt)
        elif kind in ("add", "multiply"):
            sa, sb = shape[ins[0]], shape[ins[1]]
            s = sa if int(np.prod(sa)) >= int(np.prod(sb)) else sb
        elif kind in ("batch_norm_", "relu", "hardswish", "hardsigmoid", "sigmoid", "scale"):
            s = shape[ins[0]]
        elif kind == "pool2d":
            n, c, h, w = shape[ins[0]]
            if a["adaptive"]:
                s = (n, c, 1, 1)
            else:
                ks = const[ins[1]]
                sh, sw = a["strides"]
                pt, pl = a["paddings"][0], a["paddings"][1]
                if a.get("padding_algorithm") == "SAME":
                    ho, wo = same_padding(h, ks[0], sh)[1], same_padding(w, ks[1], sw)[1]
                elif a["ceil_mode"]:
                    ho, wo = -(-(h + 2 * pt - ks[0]) // sh) + 1, -(-(w + 2 * pl - ks[1]) // sw) + 1
                else:
                    ho, wo = (h + 2 * pt - ks[0]) // sh + 1, (w + 2 * pl - ks[1]) // sw + 1
                s = (n, c, ho, wo)
        elif kind == "nearest_interp":
            n, c, h, w = shape[ins[0]]
            sc = int(a["scale"][0])
            s = (n, c, h * sc, w * sc)
        elif kind == "concat":
            parts, dim = const[ins[0]], int(const[ins[1]])
            ss = [list(shape[p]) for p in parts]
            s = list(ss[0])
            s[dim] = sum(x[dim] for x in ss)
            s = tuple(s)
        else:
            raise NotImplementedError(f"detector op {kind}")
        shape[outs[0]] = s
    return shape, const


class _Compiler:
    def __init__(self, graph, params, xshape):
        self.g, self.ops, self.params = graph, graph.ops, params
        self.shape, self.const = infer_shapes(graph, xshape)
        self.plan = Plan()
        self.consumers, self.producer = {}, {}
        for i, (kind, ins, outs, a) in enumerate(self.ops):
            for v in ins:
                self.consumers.setdefault(v, []).append(i)
            for v in outs:
                self.producer[v] = i
        self.val = {}
        self.skip = set()
        self._nbuf = 0
        self.opts = _opts()
        self._analyse()

    # ---------------------------------------------------------------- analysis
    def sole_reader(self, v):
        c = self.consumers.get(v, [])
        return c[0] if len(c) == 1 and v != self.g.output_id else None

    def param_behind(self, v):
        """the parameter array a bias operand is (a reshape of), else None"""
        if v in self.params:
            return self.params[v]
        j = self.producer.get(v)
        if j is not None and self.ops[j][0] == "reshape" and self.ops[j][1][0] in self.params:
            return self.params[self.ops[j][1][0]]
        return None

    def runtime4(self, v):
        return v not in self.params and v in self.shape and len(self.shape[v]) == 4 and self.param_behind(v) is None

    def chain_after(self, vid, acts=("relu",)):
        """what rides on the producer of `vid`: [add of a per-channel parameter] -> [batch_norm_] -> [activation], each the only reader of the one before"""
        ch = dict(bias=None, bn=None, act=None, folded=[], final=vid)
        cur = vid
        j = self.sole_reader(cur)
        if j is not None and self.ops[j][0] == "add":
            i2 = self.ops[j][1]
            other = i2[1] if i2[0] == cur else i2[0]
            pb = self.param_behind(other)
            if pb is not None and pb.size == self.shape[cur][1]:
                ch["bias"] = np.asarray(pb, np.float64).reshape(-1)
                ch["folded"].append(j)
                cur = self.ops[j][2][0]
                j = self.sole_reader(cur)
        if j is not None and self.ops[j][0] == "batch_norm_" and self.ops[j][1][0] == cur:
            ch["bn"] = j
            ch["folded"].append(j)
            cur = self.ops[j][2][0]
            j = self.sole_reader(cur)
        if j is not None and self.ops[j][0] in acts:
            ch["act"] = self.ops[j][0]
            ch["folded"].append(j)
            cur = self.ops[j][2][0]
        ch["final"] = cur
        return ch

    def bn_affine64(self, j):
        kind, ins, outs, a = self.ops[j]
        mean, var, gamma, beta = (np.asarray(self.params[v], np.float64) for v in ins[1:5])
        s = gamma / np.sqrt(var + a["epsilon"])
        return s, beta - mean * s

    def conv_geometry(self, i):
        kind, ins, outs, a = self.ops[i]
        n, cin, h, w = self.shape[ins[0]]
        cout, _, kh, kw = self.shape[ins[1]]
        sh, sw = a["strides"]
        pt, pl = a["paddings"][0], a["paddings"][1]
        dil = list(a.get("dilations", [1, 1]))
        if a.get("padding_algorithm") == "SAME":
            (pt, ho), (pl, wo) = same_padding(h, kh, sh, dil[0]), same_padding(w, kw, sw, dil[1])
        else:
            ho, wo = (h + 2 * pt - kh) // sh + 1, (w + 2 * pl - kw) // sw + 1
        pb, pr = max(0, (ho - 1) * sh + kh - pt - h), max(0, (wo - 1) * sw + kw - pl - w)
        return dict(n=n, cin=cin, h=h, w=w, cout=cout, kh=kh, kw=kw, sh=sh, sw=sw, pt=pt, pl=pl, ho=ho, wo=wo, halo=(max(pt, pb), max(pl, pr)), dil=dil)

    def is_gemm_conv(self, i):
        kind, ins, outs, a = self.ops[i]
        if kind != "conv2d" or a["groups"] != 1 or not self.runtime4(ins[0]):
            return False
        ge = self.conv_geometry(i)
        # (ocr_det.run keeps K < 64 on the direct kernel because of its layout passes; a 1x1 conv of 32 channels is one K chunk here)
        return (ge["cin"] * ge["kh"] * ge["kw"] >= GEMM_MIN_K or ge["cin"] >= 16) and ge["dil"] == [1, 1] and ge["cout"] > 4

    def is_dot_conv(self, i):
        kind, ins, outs, a = self.ops[i]
        if kind != "conv2d" or a["groups"] != 1 or not self.runtime4(ins[0]):
            return False
        ge = self.conv_geometry(i)
        return (ge["cout"] == 1 and (ge["kh"], ge["kw"], ge["sh"], ge["sw"], ge["pt"], ge["pl"]) == (1, 1, 1, 1, 0, 0) and ge["cin"] % 4 == 0
                and ge["cin"] >= 32)

    def is_dw_view(self, i):
        kind, ins, outs, a = self.ops[i]
        if kind != "depthwise_conv2d" or not self.runtime4(ins[0]):
            return False
        cin = self.shape[ins[0]][1]
        return a["groups"] == cin and self.shape[ins[1]][0] == cin and cin % 4 == 0 and list(a.get("dilations", [1, 1])) == [1, 1]

    def deconv_kind(self, i):
        kind, ins, outs, a = self.ops[i]
        if kind != "conv2d_transpose" or not self.runtime4(ins[0]):
            return None
        n, cin, h, w = self.shape[ins[0]]
        wshape = self.shape[ins[1]]
        if tuple(wshape[2:]) != (2, 2) or list(a["strides"]) != [2, 2] or list(a["paddings"]) != [0, 0]:
            raise NotImplementedError("conv2d_transpose other than 2x2 / stride 2")
        if a["groups"] != 1:
            return "nchw"
        cout = wshape[1]
        if cin % 32 == 0 and cout % 32 == 0:
            return "gemm"
        if cout == 1 and cin % 4 == 0:
            return "dots"
        return "nchw"

    def _analyse(self):
        ops = self.ops
        # concats that become slices of one NHWC buffer: channel concats of 4-d runtime values read by at least one GEMM conv
        self.part_of, self.nhwc_concat = {}, {}
        for i, (kind, ins, outs, a) in enumerate(ops):
            if kind != "concat":
                continue
            parts, dim = self.const[ins[0]], int(self.const[ins[1]])
            if dim != 1 or not all(self.runtime4(p) for p in parts) or len(set(parts)) != len(parts):
                continue
            if not any(self.is_gemm_conv(j) for j in self.consumers.get(outs[0], [])):
                continue
            offs, at = [], 0
            for p in parts:
                offs.append(at)
                at += r32(self.shape[p][1])
            self.nhwc_concat[outs[0]] = dict(parts=list(parts), offs=offs, Cs=at, op=i, im2col={})
            # a part of few channels read by ONE stride-1 conv: its slice holds that conv's neighbourhoods of it (one K chunk instead of
            # one per tap: the 65-channel 3x3 conv of the DB head is 19 chunks per pixel instead of 27)
            readers = self.consumers.get(outs[0], [])
            if self.opts["im2col"] and len(readers) == 1 and self.is_gemm_conv(readers[0]) and outs[0] != self.g.output_id:
                ge = self.conv_geometry(readers[0])
                for k, p in enumerate(parts):
                    if (ge["sh"], ge["sw"]) == (1, 1) and self.shape[p][1] * ge["kh"] * ge["kw"] <= 32 and ge["kh"] * ge["kw"] > 1:
                        self.nhwc_concat[outs[0]]["im2col"][k] = (ge["kh"], ge["kw"], ge["pt"], ge["pl"])
            for k, p in enumerate(parts):
                self.part_of.setdefault(p, (outs[0], k))
        # halo every value needs: the largest padding of the convs that read it
        self.need = {}
        for i, (kind, ins, outs, a) in enumerate(ops):
            if kind in ("conv2d", "depthwise_conv2d") and self.runtime4(ins[0]):
                hh, hw = self.conv_geometry(i)["halo"]
                o = self.need.get(ins[0], (0, 0))
                self.need[ins[0]] = (max(o[0], hh), max(o[1], hw))
        # chains riding on GEMM producers, and which of their results wait for an add
        self.chain, self.gemm_final = {}, {}
        for i, (kind, ins, outs, a) in enumerate(ops):
            if self.is_gemm_conv(i) or self.deconv_kind(i) == "gemm":
                ch = self.chain_after(outs[0], acts=("relu",))
                self.chain[i] = ch
                self.gemm_final[ch["final"]] = i
            elif self.is_dot_conv(i) or self.deconv_kind(i) == "dots":
                self.chain[i] = self.chain_after(outs[0], acts=("relu", "hardswish", "sigmoid"))
            elif self.is_dw_view(i):
                ch = self.chain_after(outs[0], acts=("relu", "hardswish"))
                if ch["bias"] is not None:                     # (no program has a bias add behind a depthwise conv: keep it a separate op)
                    ch = dict(bias=None, bn=None, act=None, folded=[], final=outs[0])
                self.chain[i] = ch

    def residual_add(self, j):
        """op j is an add of two same-shape 4-d runtime values"""
        kind, ins, outs, a = self.ops[j]
        return (kind == "add" and len(ins) == 2 and self.runtime4(ins[0]) and self.runtime4(ins[1]) and self.shape[ins[0]] == self.shape[ins[1]]
                and ins[0] != ins[1])

    def deferrable(self, final_vid):
        """the GEMM result `final_vid` is only read by a same-shape add: the GEMM waits for that add and takes the other operand as its
        residual.  When both operands are such results the LATER conv carries the residual and the earlier one runs in program order
        (the four 9x9 convs of the neck then stay adjacent and independent: one launch, group_gemms)"""
        j = self.sole_reader(final_vid)
        if j is None or not self.residual_add(j) or final_vid not in self.gemm_final:
            return False
        i2 = self.ops[j][1]
        other = i2[1] if i2[0] == final_vid else i2[0]
        if other in self.gemm_final and self.sole_reader(other) == j and self.gemm_final[other] > self.gemm_final[final_vid]:
            return False
        return True

    def lazy_upsample(self, i):
        """nearest_interp i only feeds an add whose other operand is a deferred GEMM result: it becomes that GEMM's residual rows"""
        kind, ins, outs, a = self.ops[i]
        j = self.sole_reader(outs[0])
        if j is None or not self.residual_add(j) or not self.runtime4(ins[0]):
            return False
        i2 = self.ops[j][1]
        other = i2[1] if i2[0] == outs[0] else i2[0]
        return self.deferrable(other) and self.shape[ins[0]][1] % 4 == 0

    # ---------------------------------------------------------------- buffers and views
    def new_buffer(self, name, size, zero):
        assert name not in self.plan.buffers, name
        self.plan.buffers[name] = [int(size), bool(zero)]
        return name

    def add_const(self, name, arr):
        assert name not in self.plan.consts, name
        self.plan.consts[name] = np.ascontiguousarray(arr)
        return name

    def halo_of(self, vid):
        return self.need.get(vid, (0, 0))

    def home(self, vid):
        """the NHWC view value `vid` is produced into: its slice of a concat buffer, or a buffer of its own"""
        n, C, H, W = self.shape[vid]
        if vid in self.part_of:
            cv, k = self.part_of[vid]
            cc = self.nhwc_concat[cv]
            if "buf" not in cc:
                ph, pw = self.halo_of(cv)
                for p in cc["parts"]:
                    if self.part_of[p][0] == cv:
                        ph, pw = max(ph, self.halo_of(p)[0]), max(pw, self.halo_of(p)[1])
                cc["halo"] = (ph, pw)
                cc["buf"] = self.new_buffer(f"cat{cv}", n * (H + 2 * ph) * (W + 2 * pw) * cc["Cs"], True)
            ph, pw = cc["halo"]
            return View(cc["buf"], n, H, W, C, ph, pw, cc["Cs"], cc["offs"][k])
        ph, pw = self.halo_of(vid)
        cs = r32(C)
        return View(self.new_buffer(f"v{vid}", n * (H + 2 * ph) * (W + 2 * pw) * cs, True), n, H, W, C, ph, pw, cs)

    def ensure_view(self, v, ph=0, pw=0, contiguous=False, dst=None):
        """an NHWC view of v with a halo of at least (ph, pw) (and one run of channels if `contiguous`); emits a to_view when v only has
        NCHW planes.  dst: write into this view instead of a fresh buffer"""
        vw = v.view
        if dst is None and vw is not None and vw.ph >= ph and vw.pw >= pw and not (contiguous and vw.gapped):
            return vw
        src = self.as_nchw(v)
        n, C, H, W = v.shape
        if dst is None:
            nh = self.halo_of(v.vid) if v.vid is not None else (0, 0)
            ph, pw = max(ph, nh[0]), max(pw, nh[1])
            self._nbuf += 1
            dst = View(self.new_buffer(f"t{self._nbuf}_{v.vid}", n * (H + 2 * ph) * (W + 2 * pw) * r32(C), True), n, H, W, C, ph, pw, r32(C))
        self.plan.steps.append(("to_view", dict(x=src, n=n, C=C, H=H, W=W, Cw=r32(C), out=dst.buf, out_off=dst.origin, img_stride=dst.img_stride,
                                                row_stride=dst.row_stride, Cs=dst.Cs)))
        if v.view is None or (v.view.ph < ph or v.view.pw < pw) or (contiguous and v.view.gapped):
            v.view = dst
        return dst

    def as_nchw(self, v):
        if v.nchw is not None:
            return v.nchw
        assert v.view is not None and v.deferred is None and v.lazy_up is None, "value has no storage"
        n, C, H, W = v.shape
        self._nbuf += 1
        name = self.new_buffer(f"p{self._nbuf}_{v.vid}", n * C * H * W, False)
        vw = v.view
        for ls, cnt, ps in vw.cmap:
            self.plan.steps.append(("from_view", dict(inp=vw.buf, in_off=vw.origin + ps, img_stride=vw.img_stride, row_stride=vw.row_stride, Cs=vw.Cs,
                                                      n=n, C=cnt, H=H, W=W, out=name, out_off=ls * H * W, out_img_stride=C * H * W)))
        v.nchw = name
        return name

    def new_nchw(self, vid, shape):
        self._nbuf += 1
        return self.new_buffer(f"o{self._nbuf}_{vid}", int(np.prod(shape)), False)

    # ---------------------------------------------------------------- GEMM producers
    def fold_affine(self, ch, cout):
        """(per-output-channel weight scale, bias, GEMM activation) of a chain, float64"""
        scale, shift = np.ones(cout, np.float64), np.zeros(cout, np.float64)
        if ch["bias"] is not None:
            shift = shift + ch["bias"]
        if ch["bn"] is not None:
            s, t = self.bn_affine64(ch["bn"])
            scale, shift = scale * s, shift * s + t
        has_bias = ch["bias"] is not None or ch["bn"] is not None
        return scale, shift, has_bias

    def emit_gemm(self, tag, src, wpacked, bias, M, N, K, row_a, col_a, row_c, col_c, dst_buf, act, res=None, row_r=None):
        bm, bn, cfg = (128, 128, TILE_128x128) if N >= 128 else (128, 64, TILE_128x64)
        if N == 32 and self.opts["n32"]:
            bm, bn, cfg = 256, 32, TILE_256x32
        tiles_m, tiles_n = -(-M // bm), -(-N // bn)
        variant = thin_variant(M, N, K, tiles_m * tiles_n) if self.opts["thin"] else 3

        def pad(a, size, fill):
            out = np.full(size, fill, np.int64)
            out[:len(a)] = a
            return out

        tabs = dict(rowA=pad(row_a, tiles_m * bm, row_a[0]), colA=np.asarray(col_a, np.int64), rowB=pad(np.arange(N, dtype=np.int64) * K, tiles_n * bn, 0),
                    colB=np.arange(K // 32, dtype=np.int64) * 32, rowC=pad(row_c, tiles_m * bm, 0), colC=pad(col_c, tiles_n * bn // 32, 0))
        if row_r is not None:
            tabs["rowR"] = pad(row_r, tiles_m * bm, 0)
        for k, t in tabs.items():
            if len(t) and (int(t.max()) + 4096 >= 2 ** 31 or int(t.min()) < 0):
                raise ValueError(f"{tag}: offset table {k} does not fit 32 bits")
        names = {k: self.add_const(f"{tag}.{k}", t.astype(np.int32)) for k, t in tabs.items()}
        wname = self.add_const(f"{tag}.w", np.asarray(wpacked, np.float32).reshape(-1))
        bname = None
        if bias is not None:
            bpad = np.zeros(tiles_n * bn, np.float32)
            bpad[:N] = bias
            bname = self.add_const(f"{tag}.bias", bpad)
        self.plan.steps.append(("gemm", dict(tag=tag, A=src, B=wname, C=dst_buf, bias=bname, R=res, tables=names, M=M, N=N, K=K, tiles_m=tiles_m, tiles_n=tiles_n,
                                             tile_cfg=cfg, act=act, variant=variant, group=None)))

    def residual_tables(self, res, ho, wo, n):
        """(buffer, rowR) of a residual operand: a view of the output's size, or (view, s) = a nearest_interp of a smaller one"""
        view, s = res
        img, pix = np.divmod(np.arange(n * ho * wo, dtype=np.int64), ho * wo)
        oy, ox = np.divmod(pix, wo)
        return view.buf, view.pix(img, oy // s, ox // s)

    def emit_conv(self, i, res=None, final=None):
        """dense conv i on the gather-GEMM, NHWC in and out, chain folded, optional residual (then `final` = the add's value); returns the output view"""
        kind, ins, outs, a = self.ops[i]
        ge, ch = self.conv_geometry(i), self.chain[i]
        xin = self.val[ins[0]]
        src = self.ensure_view(xin, *ge["halo"])
        n, cin, cout, kh, kw, sh, sw, pt, pl, ho, wo = (ge[k] for k in ("n", "cin", "cout", "kh", "kw", "sh", "sw", "pt", "pl", "ho", "wo"))
        dst = self.home(ch["final"] if final is None else final)
        npad = r32(cout)
        scale, shift, has_bias = self.fold_affine(ch, cout)
        w64 = np.asarray(self.params[ins[1]], np.float64) * scale[:, None, None, None]
        # K = (tap, 32-float chunk of the physical channels) [+ one chunk per im2col part, read at the output pixel itself]
        im = src.im2col or {}
        regular = [(ls, cnt, ps) for k, (ls, cnt, ps) in enumerate(src.cmap) if k not in im]
        chunks = sorted({ps + 32 * j for ls, cnt, ps in regular for j in range(r32(cnt) // 32)})
        nch = len(chunks)
        where = {c: k for k, c in enumerate(chunks)}
        wp = np.zeros((npad, kh * kw * nch + len(im), 32), np.float32)
        for ls, cnt, ps in regular:
            wseg = w64[:, ls:ls + cnt].transpose(0, 2, 3, 1).reshape(cout, kh * kw, cnt).astype(np.float32)
            for j in range(0, cnt, 32):
                m = min(32, cnt - j)
                wp[:cout, where[ps + j]::nch, :m][:, :kh * kw] = wseg[:, :, j:j + m]
        col_a = [((t // kw) * src.Wp + t % kw) * src.Cs + c for t in range(kh * kw) for c in chunks]
        for e, (k, geo) in enumerate(sorted(im.items())):
            ls, cnt, ps = src.cmap[k]
            assert geo == (kh, kw, pt, pl) and (sh, sw) == (1, 1) and cnt * kh * kw <= 32
            wp[:cout, kh * kw * nch + e, :cnt * kh * kw] = w64[:, ls:ls + cnt].reshape(cout, cnt * kh * kw).astype(np.float32)
            col_a.append((pt * src.Wp + pl) * src.Cs + ps)
        M, K = n * ho * wo, wp.shape[1] * 32
        img, pix = np.divmod(np.arange(M, dtype=np.int64), ho * wo)
        oy, ox = np.divmod(pix, wo)
        row_a = src.pix(img, oy * sh - pt, ox * sw - pl)
        col_a = np.asarray(col_a, np.int64)
        row_c = dst.pix(img, oy, ox)
        bias = None
        if has_bias:
            bias = np.zeros(npad, np.float32)
            bias[:cout] = shift.astype(np.float32)
        rbuf = row_r = None
        if res is not None:
            rbuf, row_r = self.residual_tables(res, ho, wo, n)
        self.emit_gemm(f"conv{i}", src.buf, wp, bias, M, npad, K, row_a, col_a, row_c, np.arange(npad // 32, dtype=np.int64) * 32, dst.buf,
                       GEMM_ACT_RELU if ch["act"] == "relu" else GEMM_ACT_NONE, rbuf, row_r)
        self.plan.flops += 2.0 * n * cout * ho * wo * kh * kw * cin
        return dst

    def emit_deconv(self, i, res=None, final=None):
        """conv2d_transpose 2x2 / stride 2 as one GEMM: rows = input pixels, columns = (dy, dx, cout), the C tables scatter the taps"""
        kind, ins, outs, a = self.ops[i]
        ch = self.chain[i]
        xin = self.val[ins[0]]
        src = self.ensure_view(xin, 0, 0)
        n, cin, h, wd = xin.shape
        w = np.asarray(self.params[ins[1]], np.float64)            # [cin][cout][2][2]
        cout = w.shape[1]
        dst = self.home(ch["final"] if final is None else final)
        scale, shift, has_bias = self.fold_affine(ch, cout)
        cp = src.cphys
        wn = (w * scale[None, :, None, None]).transpose(2, 3, 1, 0).reshape(4 * cout, cin)      # B[(dy, dx, co)][ci]
        wp = np.zeros((4 * cout, cp), np.float32)
        for ls, cnt, ps in src.cmap:
            wp[:, ps:ps + cnt] = wn[:, ls:ls + cnt].astype(np.float32)
        M, N, K = n * h * wd, 4 * cout, cp
        img, pix = np.divmod(np.arange(M, dtype=np.int64), h * wd)
        y, x = np.divmod(pix, wd)
        row_a = src.pix(img, y, x)
        row_c = dst.pix(img, 2 * y, 2 * x)
        tap, cc = np.divmod(np.arange(N // 32, dtype=np.int64), cout // 32)
        col_c = ((tap // 2) * dst.Wp + tap % 2) * dst.Cs + cc * 32
        bias = np.tile(shift.astype(np.float32), 4) if has_bias else None
        assert res is None, "a transposed conv as the deferred operand of an add is not planned"
        self.emit_gemm(f"deconv{i}", src.buf, wp, bias, M, N, K, row_a, np.arange(K // 32, dtype=np.int64) * 32, row_c, col_c, dst.buf,
                       GEMM_ACT_RELU if ch["act"] == "relu" else GEMM_ACT_NONE)
        self.plan.flops += 2.0 * n * h * wd * 4 * cout * cin
        return dst

    def materialize(self, v, res=None):
        """run the GEMM a value was waiting with"""
        i = v.deferred
        v.deferred = None
        v.view = self.emit_conv(i, res) if self.ops[i][0] == "conv2d" else self.emit_deconv(i, res)
        return v.view

    def settle(self, v):
        """a value somebody other than its planned reader wants: give it storage now"""
        if v.deferred is not None:
            self.materialize(v)
        if v.lazy_up is not None:
            src, s = v.lazy_up
            v.lazy_up = None
            self.emit_nearest(src, s, v)
        return v

    def emit_nearest(self, src, s, out):
        n, C, H, W = out.shape
        sv = self.ensure_view(src, 0, 0, contiguous=True)
        dst = self.home(out.vid)
        self.plan.steps.append(("nearest_view", dict(inp=sv.buf, in_off=sv.origin, in_img=sv.img_stride, in_row=sv.row_stride, in_cs=sv.Cs, n=n, C=C, Ho=H,
                                                     Wo=W, s=s, out=dst.buf, out_off=dst.origin, out_img=dst.img_stride, out_row=dst.row_stride,
                                                     out_cs=dst.Cs)))
        out.view = dst

    # ---------------------------------------------------------------- the walk
    def get(self, vid):
        return self.settle(self.val[vid])

    def compile(self):
        g, plan = self.g, self.plan
        for v, arr in self.params.items():
            self.val[v] = Val(arr.shape, const=arr, vid=v)
        for v, c in self.const.items():
            self.val[v] = Val(const=c, vid=v)
        xs = self.shape[g.input_id]
        plan.input = self.new_buffer("x", int(np.prod(xs)), False)
        self.val[g.input_id] = Val(xs, nchw="x", vid=g.input_id)
        counts = {}
        for i, (kind, ins, outs, a) in enumerate(self.ops):
            if i in self.skip or kind in ("full_int_array", "full", "combine"):
                continue
            how = self.op(i, kind, ins, outs, a)
            counts[how] = counts.get(how, 0) + 1
        out = self.get(g.output_id)
        plan.output = (self.as_nchw(out), out.shape)
        plan.stats = counts
        if self.opts["group"]:
            group_gemms(plan)
        return plan

    def finish(self, i, final, v):
        """register the result of producer i whose chain ends in value `final`"""
        v.vid = final
        self.val[final] = v
        self.skip.update(self.chain[i]["folded"])

    def op(self, i, kind, ins, outs, a):
        S, plan = self.shape, self.plan
        if kind == "conv2d" and self.is_gemm_conv(i):
            final = self.chain[i]["final"]
            v = Val(S[final], vid=final)
            if self.deferrable(final):
                v.deferred = i
                self.get(ins[0])                       # (its input is settled now, in program order)
            else:
                self.get(ins[0])
                v.view = self.emit_conv(i)
            self.finish(i, final, v)
            return "gemm conv"
        if kind == "conv2d_transpose" and self.deconv_kind(i) == "gemm":
            final = self.chain[i]["final"]
            self.get(ins[0])
            v = Val(S[final], vid=final)
            v.view = self.emit_deconv(i)
            self.finish(i, final, v)
            return "gemm deconv"
        if (kind == "conv2d" and self.is_dot_conv(i)) or (kind == "conv2d_transpose" and self.deconv_kind(i) == "dots"):
            ch = self.chain[i]
            xin = self.get(ins[0])
            sv = self.ensure_view(xin, 0, 0, contiguous=True)
            n, C, H, W = xin.shape
            w = np.asarray(self.params[ins[1]], np.float32)
            if kind == "conv2d":
                n_out, wd = 1, w.reshape(1, C)
            else:
                n_out, wd = 4, np.ascontiguousarray(w[:, 0].reshape(C, 4).T)          # [(dy, dx)][ci]
            scale = shift = None
            if ch["bn"] is not None:                   # (a batch_norm_ on one channel: fold it)
                s, t = self.bn_affine64(ch["bn"])
                wd = (wd.astype(np.float64) * s[0]).astype(np.float32)
                shift = (0.0 if ch["bias"] is None else ch["bias"][0]) * s[0] + t[0]
            elif ch["bias"] is not None:
                shift = ch["bias"][0]
            final = ch["final"]
            out = self.new_nchw(final, S[final])
            tag = f"dots{i}"
            self.plan.steps.append(("dots_view", dict(inp=sv.buf, in_off=sv.origin, in_img=sv.img_stride, in_row=sv.row_stride, in_cs=sv.Cs, n=n, C=C, H=H,
                                                      W=W, w=self.add_const(f"{tag}.w", wd.reshape(-1)),
                                                      bias=self.add_const(f"{tag}.b", np.array([shift], np.float32)) if shift is not None else None,
                                                      n_out=n_out, act=ACT_CODES[ch["act"]], out=out)))
            plan.flops += 2.0 * n * H * W * C * n_out
            self.finish(i, final, Val(S[final], nchw=out))
            return "dots"
        if kind == "depthwise_conv2d" and self.is_dw_view(i):
            ch, ge = self.chain[i], self.conv_geometry(i)
            xin = self.get(ins[0])
            sv = self.ensure_view(xin, *ge["halo"], contiguous=True)
            final = ch["final"]
            dst = self.home(final)
            C = ge["cin"]
            w = np.asarray(self.params[ins[1]], np.float32).reshape(C, ge["kh"] * ge["kw"])
            tag = f"dw{i}"
            sc = sh_ = None
            if ch["bn"] is not None:
                s, t = self.bn_affine64(ch["bn"])
                sc, sh_ = self.add_const(f"{tag}.scale", s.astype(np.float32)), self.add_const(f"{tag}.shift", t.astype(np.float32))
            self.plan.steps.append(("dwconv_view", dict(inp=sv.buf, in_off=sv.origin, in_img=sv.img_stride, in_row=sv.row_stride, in_cs=sv.Cs,
                                                        w=self.add_const(f"{tag}.w", np.ascontiguousarray(w.T).reshape(-1)), scale=sc, shift=sh_, n=ge["n"], C=C,
                                                        kh=ge["kh"], kw=ge["kw"], sh=ge["sh"], sw=ge["sw"], pt=ge["pt"], pl=ge["pl"], Ho=ge["ho"], Wo=ge["wo"],
                                                        act=ACT_CODES[ch["act"]], out=dst.buf, out_off=dst.origin, out_img=dst.img_stride,
                                                        out_row=dst.row_stride, out_cs=dst.Cs)))
            plan.flops += 2.0 * ge["n"] * C * ge["ho"] * ge["wo"] * ge["kh"] * ge["kw"]
            self.finish(i, final, Val(S[final], view=dst))
            return "dw view"
        if kind == "nearest_interp":
            s = int(a["scale"][0])
            if a["scale"][0] != a["scale"][1] or s != a["scale"][0]:
                raise NotImplementedError("non-integer nearest_interp scale")
            xin = self.get(ins[0])
            out = Val(S[outs[0]], vid=outs[0])
            self.val[outs[0]] = out
            if self.lazy_upsample(i):
                out.lazy_up = (xin, s)
                return "nearest as residual rows"
            if xin.view is not None and xin.shape[1] % 4 == 0:
                self.emit_nearest(xin, s, out)
                return "nearest view"
            n, c, h, w = xin.shape
            out.nchw = self.new_nchw(outs[0], out.shape)
            plan.steps.append(("nearest_nchw", dict(x=self.as_nchw(xin), planes=n * c, H=h, W=w, s=s, out=out.nchw)))
            return "nearest nchw"
        if kind == "add" and self.residual_add(i):
            va, vb = self.val[ins[0]], self.val[ins[1]]
            if va.deferred is not None or vb.deferred is not None:
                if va.deferred is None:
                    va, vb = vb, va                         # va: the GEMM that takes the other operand as its residual
                if vb.deferred is not None:
                    self.materialize(vb)
                if vb.lazy_up is not None:
                    src, s = vb.lazy_up
                    vb.lazy_up = None
                    res = (self.ensure_view(src, 0, 0, contiguous=True), s)
                else:
                    res = (self.ensure_view(vb, 0, 0, contiguous=True), 1)
                conv = va.deferred
                va.deferred = None                          # the GEMM writes the add's value: its chain result was only ever read here
                view = self.emit_conv(conv, res, outs[0]) if self.ops[conv][0] == "conv2d" else self.emit_deconv(conv, res, outs[0])
                self.val[outs[0]] = Val(S[outs[0]], view=view, vid=outs[0])
                return "add as residual"
        if kind == "concat" and outs[0] in self.nhwc_concat:
            cc = self.nhwc_concat[outs[0]]
            n, C, H, W = S[outs[0]]
            cmap, at, first = [], 0, None
            for k, p in enumerate(cc["parts"]):
                v = self.get(p)
                slice_view = None
                if k in cc["im2col"]:
                    save_part = self.part_of.get(p)
                    self.part_of[p] = (outs[0], k)
                    slice_view = self.home(p)
                    if save_part is not None:
                        self.part_of[p] = save_part
                    kh_, kw_, pt_, pl_ = cc["im2col"][k]
                    pn, pc, ph_, pw_ = v.shape
                    self.plan.steps.append(("im2col_view", dict(x=self.as_nchw(v), n=pn, C=pc, H=ph_, W=pw_, kh=kh_, kw=kw_, pt=pt_, pl=pl_, out=slice_view.buf,
                                                                out_off=slice_view.origin, out_img=slice_view.img_stride, out_row=slice_view.row_stride,
                                                                out_cs=slice_view.Cs)))
                    first = first or slice_view
                    cmap.append((at, S[p][1], cc["offs"][k]))
                    at += S[p][1]
                    continue
                if self.part_of[p] == (outs[0], k):
                    if v.view is not None and v.view.buf == cc.get("buf") and v.view.c0 == cc["offs"][k]:
                        slice_view = v.view
                if slice_view is None:                      # produced elsewhere (NCHW kernels, another concat): copy it into its slice
                    save_part = self.part_of.get(p)
                    self.part_of[p] = (outs[0], k)
                    slice_view = self.home(p)
                    if save_part is not None:
                        self.part_of[p] = save_part
                    keep = v.view
                    self.ensure_view(v, dst=slice_view)
                    if keep is not None:
                        v.view = keep
                first = first or slice_view
                cmap.append((at, S[p][1], cc["offs"][k]))
                at += S[p][1]
            cview = View(first.buf, n, H, W, C, first.ph, first.pw, cc["Cs"], 0, cmap)
            if cc["im2col"]:
                cview.im2col = dict(cc["im2col"])
            self.val[outs[0]] = Val(S[outs[0]], view=cview, vid=outs[0])
            return "concat as slices"
        return self.nchw_op(i, kind, ins, outs, a)

    # ---------------------------------------------------------------- operators that stay on the NCHW kernels
    def nchw_op(self, i, kind, ins, outs, a):
        S, plan, steps = self.shape, self.plan, self.plan.steps
        g = lambda j: self.get(ins[j])

        def out_buf():
            v = Val(S[outs[0]], nchw=self.new_nchw(outs[0], S[outs[0]]), vid=outs[0])
            self.val[outs[0]] = v
            return v.nchw

        if kind in ("conv2d", "depthwise_conv2d"):
            ge = self.conv_geometry(i)
            xin = g(0)
            cin = ge["cin"]
            dw = 1 if a["groups"] == cin and a["groups"] > 1 else 0
            if not dw and a["groups"] != 1:
                raise NotImplementedError("grouped conv")
            if ge["dil"] != [1, 1]:
                raise NotImplementedError("dilated conv")
            x = self.as_nchw(xin)
            steps.append(("conv_nchw", dict(x=x, w=self.weight_const(ins[1]), n=ge["n"], cin=cin, h=ge["h"], wd=ge["w"], cout=ge["cout"], kh=ge["kh"],
                                            kw=ge["kw"], sh=ge["sh"], sw=ge["sw"], pt=ge["pt"], pl=ge["pl"], ho=ge["ho"], wo=ge["wo"], dw=dw, out=out_buf())))
            plan.flops += 2.0 * ge["n"] * ge["cout"] * ge["ho"] * ge["wo"] * ge["kh"] * ge["kw"] * (1 if dw else cin)
            return "conv nchw"
        if kind == "conv2d_transpose":
            xin = g(0)
            n, cin, h, wd = xin.shape
            dw = 1 if a["groups"] == cin and a["groups"] > 1 else 0
            cout = cin if dw else S[ins[1]][1]
            x = self.as_nchw(xin)
            steps.append(("deconv_nchw", dict(x=x, w=self.weight_const(ins[1]), n=n, cin=cin, h=h, wd=wd, cout=cout, dw=dw, out=out_buf())))
            plan.flops += 2.0 * n * h * wd * 4 * cout * (1 if dw else cin)
            return "deconv nchw"
        if kind == "batch_norm_":
            xin = g(0)
            s, t = self.bn_affine64(i)
            x = self.as_nchw(xin)
            steps.append(("affine", dict(x=x, scale=self.add_const(f"bn{i}.scale", s.astype(np.float32)), shift=self.add_const(f"bn{i}.shift", t.astype(np.float32)),
                                         total=int(np.prod(xin.shape)), C=xin.shape[1], HW=xin.shape[2] * xin.shape[3], out=out_buf())))
            return "affine"
        if kind == "reshape":
            xin = g(0)
            tgt = S[outs[0]]
            if xin.const is not None:
                self.val[outs[0]] = Val(tgt, const=np.asarray(xin.const).reshape(tgt), vid=outs[0])
            else:
                self.val[outs[0]] = Val(tgt, nchw=self.as_nchw(xin), vid=outs[0])
            return "reshape"
        if kind in ("add", "multiply"):
            va, vb = g(0), g(1)
            numel = lambda v: int(np.asarray(v.const).size) if v.const is not None else int(np.prod(v.shape))
            if va.const is not None and vb.const is None or (va.const is None and vb.const is None and numel(va) < numel(vb)):
                va, vb = vb, va                             # add / multiply commute: keep the full tensor first
            if va.const is not None:
                raise NotImplementedError("binary op on two constants")
            n, c = va.shape[0], va.shape[1]
            total = numel(va)
            hw = total // (n * c)
            nb = numel(vb)
            mode = 0 if (vb.const is None and tuple(vb.shape) == tuple(va.shape)) else 3 if nb == 1 else 1 if nb == c else 2 if nb == n * c else None
            if mode is None:
                raise NotImplementedError(f"broadcast {va.shape} with {vb.shape}")
            xa = self.as_nchw(va)
            if vb.const is not None:
                b = ("const", self.add_const(f"op{i}.b", np.asarray(vb.const, np.float32).reshape(-1)))
            else:
                b = ("buf", self.as_nchw(vb))
            steps.append(("binary", dict(a=xa, b=b, op=0 if kind == "add" else 1, total=total, C=c, HW=hw, mode=mode, out=out_buf())))
            return "binary"
        if kind in ("relu", "hardswish", "hardsigmoid", "sigmoid", "scale"):
            xin = g(0)
            code, p0, p1 = {"relu": 0, "hardswish": 1, "hardsigmoid": 2, "sigmoid": 3, "scale": 4}[kind], 0.0, 0.0
            if kind == "hardsigmoid":
                p0, p1 = a["slope"], a["offset"]
            elif kind == "scale":
                sv = self.val[ins[1]].const if len(ins) > 1 and ins[1] in self.val and self.val[ins[1]].const is not None else a.get("scale", 1.0)
                p0, b = float(np.asarray(sv).reshape(-1)[0]) if not isinstance(sv, (int, float)) else float(sv), float(a.get("bias", 0.0))
                p1 = b if a.get("bias_after_scale", True) else b * p0
            x = self.as_nchw(xin)
            steps.append(("unary", dict(x=x, total=int(np.prod(xin.shape)), kind=code, p0=float(p0), p1=float(p1), out=out_buf())))
            return "unary"
        if kind == "pool2d":
            xin, ks = g(0), self.val[ins[1]].const
            n, c, h, wd = xin.shape
            x = self.as_nchw(xin)
            if a["adaptive"]:
                if list(ks) != [1, 1] or a["pooling_type"] != "avg":
                    raise NotImplementedError("adaptive pool other than global average")
                steps.append(("gap", dict(x=x, planes=n * c, HW=h * wd, out=out_buf())))
                return "gap"
            if a["pooling_type"] != "max":
                raise NotImplementedError("average pool")
            sh, sw = a["strides"]
            pt, pl = a["paddings"][0], a["paddings"][1]
            if a.get("padding_algorithm") == "SAME":
                pt, pl = same_padding(h, ks[0], sh)[0], same_padding(wd, ks[1], sw)[0]
            _, _, ho, wo = S[outs[0]]
            steps.append(("maxpool", dict(x=x, planes=n * c, H=h, W=wd, kh=ks[0], kw=ks[1], sh=sh, sw=sw, pt=pt, pl=pl, Ho=ho, Wo=wo, out=out_buf())))
            return "maxpool"
        if kind == "concat":
            parts, dim = [self.get(p) for p in self.val[ins[0]].const], int(self.val[ins[1]].const)
            if dim != 1:
                raise NotImplementedError("concat along an axis other than the channels")
            srcs = [self.as_nchw(p) for p in parts]
            out = out_buf()
            nb = S[outs[0]][0]
            pitch, at = int(np.prod(S[outs[0]])) // nb, 0
            for p, sname in zip(parts, srcs):
                w_ = int(np.prod(p.shape)) // nb
                steps.append(("copy", dict(src=sname, src_pitch=w_, dst=out, dst_off=at, dst_pitch=pitch, width=w_, rows=nb)))
                at += w_
            return "concat nchw"
        raise NotImplementedError(f"detector op {kind}")

    def weight_const(self, vid):
        name = f"param{vid}"
        if name not in self.plan.consts:
            self.add_const(name, np.asarray(self.params[vid], np.float32).reshape(-1))
        return name


def group_gemms(plan):
    """adjacent GEMM steps of one tile configuration and kernel variant that share nothing (no step reads or accumulates into a buffer
    another one writes, all write different buffers) get one group id: the runner makes them ONE resident launch list (vsr_gemm_plan_create
    takes several problems), so the few tiles of a coarse FPN level fill the last round of a fine one instead of a launch of their own"""
    gid, cur = 0, []

    def close():
        nonlocal gid, cur
        if len(cur) > 1:
            for p in cur:
                p["group"] = gid
            gid += 1
        cur = []

    for kind, p in plan.steps:
        if kind != "gemm":
            close()
            continue
        if cur:
            written = {q["C"] for q in cur}
            read = {q["A"] for q in cur} | {q["R"] for q in cur if q["R"] is not None}
            same = cur[0]["tile_cfg"] == p["tile_cfg"] and cur[0]["variant"] == p["variant"]
            if not same or p["C"] in written or p["C"] in read or p["A"] in written or (p["R"] is not None and p["R"] in written):
                close()
        cur.append(p)
    close()
    plan.stats["gemm groups"] = gid


def compile_plan(graph, params, xshape):
    """graph: paddle_graph.Graph; params: {value id: numpy array}; xshape: (n, 3, H, W) -> Plan"""
    return _Compiler(graph, {v: np.asarray(p) for v, p in params.items()}, tuple(int(s) for s in xshape)).compile()
