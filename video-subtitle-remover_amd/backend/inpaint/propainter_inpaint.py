"""ProPainter plugin with the reference's signature, running on the MI355X engines.

Mirrors backend/inpaint/propainter_inpaint.py:
  PropainterInpaint(device, model_dir, sub_video_length=80, use_fp16=True)   :139-188
      __call__(input_frames, input_mask) -> frames                            :363-418  (the generic plugin contract)
      inpaint(frames, mask) -> comp frames                                    :190-361
  read_mask :32-77 (numpy-mask branch), get_ref_index :122-136
All network arithmetic happens in libvsr_hip.so (RAFT, flow completion, image propagation, generator: vsr_raft_* /
vsr_rfc_* / vsr_pp_*), in exact fp32 by default; `precision="f16"` is the reference's own GPU arithmetic (flow completion and generator on
fp16 operands with fp32 accumulation, RAFT fp32) -- `use_fp16` itself is accepted for signature compatibility.  This file is the
host loop: mask dilation (scipy, once per batch, as in the reference), sub-video / neighbour / reference schedules.  The frames
stay in HBM as uint8 BGR from upload to download; normalise + mask, compose and the u8 blend of overlapping windows are kernels.  `model_dir` may also be a dict
{"raft": sd, "rfc": sd, "propainter": sd} of already loaded state_dicts (the shipped checkpoints are missing blobs).
"""
import contextlib
import ctypes as C
import os
import threading
import time

import numpy as np
import scipy.ndimage
import torch

from ..tools.inpaint_tools import get_inpaint_area_by_mask
from ... import switches
from ..._lib import check, lib
from ...engine import PpEngine, RaftEngine, RfcEngine
from .sttn_auto_inpaint import _device_index


def read_mask(mask, length, flow_mask_dilates=8, mask_dilates=5):
    """numpy-mask branch of the reference's read_mask: -> (flow_mask, mask_dilated) uint8 {0,1} arrays [H,W] (used for every frame)"""
    m = np.asarray(mask)
    if m.ndim == 3 and m.shape[2] == 1:
        m = m[:, :, 0]
    elif m.ndim == 3 and m.shape[2] == 3:
        # cv2.COLOR_BGR2GRAY: (B*1868 + G*9617 + R*4899 + 8192) >> 14 on uint8
        m = ((m[:, :, 0].astype(np.int32) * 1868 + m[:, :, 1].astype(np.int32) * 9617 + m[:, :, 2].astype(np.int32) * 4899 + 8192) >> 14).astype(np.uint8)

    def dil(it):
        return scipy.ndimage.binary_dilation(m, iterations=it).astype(np.uint8) if it > 0 else (m > 0.1).astype(np.uint8)

    return dil(flow_mask_dilates), dil(mask_dilates)


def encoder_cache_plan(windows, chunk=16):
    """windows: [(neighbour ids, reference ids)] of one inpaint() call.  The generator's encoder (and, for frames that are a reference
    frame of some window, its soft split) is a per-frame function: -> (calls, feat_slot, tok_slot) where calls = [(frame ids, how many
    of the first of them need tokens)] of at most `chunk` frames each and feat_slot / tok_slot map a frame to its cache entry (the
    entries are filled in call order)."""
    refs = sorted({i for _, ref in windows for i in ref})
    local = sorted({i for nb, _ in windows for i in nb} - set(refs))
    calls = [(refs[i:i + chunk], len(refs[i:i + chunk])) for i in range(0, len(refs), chunk)]
    calls += [(local[i:i + chunk], 0) for i in range(0, len(local), chunk)]
    feat_slot, tok_slot = {}, {}
    for ids, ntok in calls:
        for k, i in enumerate(ids):
            feat_slot[i] = len(feat_slot)
            if k < ntok:
                tok_slot[i] = len(tok_slot)
    return calls, feat_slot, tok_slot


def get_ref_index(mid_neighbor_id, neighbor_ids, length, ref_stride=10, ref_num=-1):
    ref_index = []
    if ref_num == -1:
        for i in range(0, length, ref_stride):
            if i not in neighbor_ids:
                ref_index.append(i)
    else:
        start_idx = max(0, mid_neighbor_id - ref_stride * (ref_num // 2))
        end_idx = min(length, mid_neighbor_id + ref_stride * (ref_num // 2))
        for i in range(start_idx, end_idx, ref_stride):
            if i not in neighbor_ids:
                if len(ref_index) > ref_num:
                    break
                ref_index.append(i)
    return ref_index


def _load(model_dir, name, file):
    """one of the three checkpoints of the mode (propainter_inpaint.py:140-146); ProPainter.pth ships in 50 MB parts and is
    assembled on first use (tools/common_tools.py, reference model_config.py:25)"""
    if isinstance(model_dir, dict):
        return model_dir[name]
    from ..tools.common_tools import checkpoint_path

    return torch.load(checkpoint_path(os.path.join(model_dir, file)), map_location="cpu")


def raft_runs(n, max_pairs, lanes=1):
    """[(first frame, end frame)] of the runs of consecutive pairs an n-frame batch goes through RAFT in: at most max_pairs // lanes
    pairs per run, as equal as possible; a run of p pairs takes p + 1 frames (neighbouring runs share one).  Pairs are independent,
    so the flows do not depend on the cut."""
    npairs = n - 1
    if npairs <= 0:
        return []
    runs = max(1, -(-npairs // max(1, max_pairs // max(1, lanes))))
    per = -(-npairs // runs)
    return [(s0, min(n, s0 + per + 1)) for s0 in range(0, npairs, per)]


class PropainterInpaint:
    accepts_device_frames = True      # __call__ also takes a uint8 [n,H,W,3] device tensor and inpaints it in place (tools/resident.py)
    # precision name -> arithmetic of (RAFT, flow completion, generator)
    PRECISIONS = {"f32": ("f32", "f32", "f32"), "f16": ("f32", "f16", "f16"), "f16-raft-split": ("split", "f16", "f16"),
                  "split": ("split", "split", "split")}

    def __init__(self, device, model_dir, sub_video_length=80, use_fp16=True, precision=None):
        self.device = device
        self.model_dir = model_dir
        # The reference halves the completion network and the generator on a GPU and keeps RAFT in fp32 (:140-146,230,249-251).
        # precision (or VSR_PP_PRECISION) picks the arithmetic of the contractions, PRECISIONS below:
        #   "f32"   (default) exact fp32 everywhere -- what the parity tests hold against the CPU oracle to the grey level;
        #   "f16"   the reference's own GPU arithmetic: RAFT exact fp32, flow completion and generator on fp16 operands with fp32
        #           accumulation (tensors, bias, activations, residuals stay fp32: no worse than `.half()` modules), range-guarded;
        #   "f16-raft-split"  the same with RAFT on fp16 hi/lo operand pairs (22 significand bits -- more than the TF32 convolutions
        #           torch gives the reference's "fp32" RAFT on a current GPU);
        #   "split" all three networks on hi/lo pairs.
        # use_fp16 (the reference's switch) does not pick the mode by itself: the default stays exact unless precision says otherwise.
        self.use_fp16 = use_fp16
        self.precision = precision or os.environ.get("VSR_PP_PRECISION", "f32")
        if self.precision not in self.PRECISIONS:
            raise ValueError(f"precision {self.precision!r}: expected one of {sorted(self.PRECISIONS)}")
        self.sub_video_length = sub_video_length
        self.neighbor_length = 10
        self.mask_dilation = 4
        self.ref_stride = 10
        self.raft_iter = 20
        self.raft_max_pairs = int(os.environ.get("VSR_RAFT_MAX_PAIRS", "35"))      # consecutive pairs per RAFT call (see inpaint())
        self.gen_lanes = int(os.environ.get("VSR_PP_LANES", "2"))                  # generator instances the sliding windows alternate over
        self.raft_lanes = int(os.environ.get("VSR_RAFT_LANES", "2"))               # RAFT instances the runs of a call alternate over (exact fp32 only)
        self._lane_models, self._lane_rafts, self._streams = {}, {}, None
        self._lane_error, self._lane_thread = None, None
        di = _device_index(device)
        self._di = di
        self.fix_raft = RaftEngine(_load(model_dir, "raft", "raft-things.pth"), device=di)
        self.fix_flow_complete = RfcEngine(_load(model_dir, "rfc", "recurrent_flow_completion.pth"), device=di)
        self.model = PpEngine(device=di, state_dict=_load(model_dir, "propainter", "ProPainter.pth"))
        self.dev = self.model.device
        # bench.py / scripts: set to a dict to have inpaint() add per-stage seconds (device-synchronised around every stage, so the
        # call gets slower) and algorithmic FLOPs: {"raft": [s, flop], "flow_completion": [...], "generator": [...], "other": [...]}
        self.profile = None
        for e, mode in zip((self.fix_raft, self.fix_flow_complete, self.model), self.PRECISIONS[self.precision]):
            if mode != "f32":
                e.set_precision(mode)
        # the lane instances pack and upload their weights on a helper thread from here on (1-2 s each: built on first use they were
        # 3 s in front of a 600-frame run's first batch, profiles/r05_e2e_pp_lanes.log); _lane_model / _lane_raft wait for the thread
        # (not a daemon, and joined at interpreter exit: a daemon thread can die in the middle of a hipMalloc / hipMemcpy when the
        # interpreter goes down before close() -- ADVICE r5.  VSR_PP_LANE_PREBUILD=0: build the lanes on first use instead.)
        if os.environ.get("VSR_PP_LANE_PREBUILD", "1") != "0":
            import atexit
            import weakref

            self._lane_thread = threading.Thread(target=self._build_lanes, name="vsr-pp-lane-build", daemon=False)
            self._lane_thread.start()
            ref = weakref.ref(self)
            atexit.register(lambda: (lambda o: o is not None and o._join_lane_thread())(ref()))

    def clone(self):
        """a second instance on the same device from the same checkpoints: its own three engines and workspaces (tools/batch_lanes.py)"""
        other = PropainterInpaint(self.device, self.model_dir, self.sub_video_length, self.use_fp16, self.precision)
        other.raft_iter = self.raft_iter
        return other

    def _new_lane_model(self):
        e = PpEngine(device=self._di, state_dict=_load(self.model_dir, "propainter", "ProPainter.pth"))
        mode = self.PRECISIONS[self.precision][2]
        if mode != "f32":
            e.set_precision(mode)
        return e

    def _new_lane_raft(self):
        return RaftEngine(_load(self.model_dir, "raft", "raft-things.pth"), device=self._di)

    def _build_lanes(self):
        try:
            if self.PRECISIONS[self.precision][0] == "f32":
                for k in range(1, max(1, self.raft_lanes)):
                    self._lane_rafts[k] = self._new_lane_raft()
            for k in range(1, max(1, self.gen_lanes)):
                self._lane_models[k] = self._new_lane_model()
        except BaseException as e:            # noqa: BLE001 -- re-raised by the first call that needs a lane
            self._lane_error = e

    def _join_lane_thread(self):
        t = self._lane_thread
        if t is not None:
            t.join()
            self._lane_thread = None

    def _lanes_built(self):
        self._join_lane_thread()
        if self._lane_error is not None:
            e, self._lane_error = self._lane_error, None
            raise e

    def _lane_model(self, k):
        """the generator instance of window lane k >= 1: same checkpoint and arithmetic, its own workspace"""
        self._lanes_built()
        if k not in self._lane_models:            # more lanes than the constructor knew of (gen_lanes set later)
            self._lane_models[k] = self._new_lane_model()
        return self._lane_models[k]

    def _lane_raft(self, k):
        """the RAFT instance of run lane k >= 1 (exact fp32: the guarded modes would need a host thread per lane)"""
        self._lanes_built()
        if k not in self._lane_rafts:
            self._lane_rafts[k] = self._new_lane_raft()
        return self._lane_rafts[k]

    def _lane_streams(self, lanes, dev):
        if self._streams is None or len(self._streams) < lanes:
            self._streams = [torch.cuda.Stream(dev) for _ in range(lanes)]
        return self._streams[:lanes]

    def close(self):
        try:
            self._lanes_built()
        except BaseException:                 # noqa: BLE001 -- closing: a lane that failed to build has nothing to release
            pass
        for e in [self.fix_raft, self.fix_flow_complete, self.model] + list(self._lane_models.values()) + list(self._lane_rafts.values()):
            e.close()
        self._lane_models, self._lane_rafts = {}, {}

    def inpaint(self, frames, mask):
        """frames: list of HxWx3 uint8 BGR crops (H, W multiples of 8), mask: HxW(x1) uint8 -> list of HxWx3 uint8 BGR.
        The batch is uploaded once as uint8 BGR and stays in HBM: RAFT reads it directly, the normalised / masked / composed
        tensors and the u8 blend of the overlapping generator windows are kernels (vsr_pp_prepare_frames / _compose_frames /
        _blend_window); one download at the end."""
        n = len(frames)
        dev = self.dev
        resident = isinstance(frames, torch.Tensor)             # a contiguous uint8 [n,h,w,3] device tensor: a device tensor comes back
        bgr = frames if resident else torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(f) for f in frames]))).to(dev)      # [n,h,w,3] BGR
        h, w = int(bgr.shape[1]), int(bgr.shape[2])
        fm, md = read_mask(mask, n, self.mask_dilation, self.mask_dilation)
        fm1, md1 = torch.from_numpy(fm).to(dev).contiguous(), torch.from_numpy(md).to(dev).contiguous()      # uint8 [h,w]
        fm_dev = fm1[None].repeat(n, 1, 1).contiguous()                                      # the engines take one mask per frame
        md_dev = md1[None].repeat(n, 1, 1).contiguous()
        P = lambda t: C.c_void_p(t.data_ptr())
        stream = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
        prof = self.profile

        def lap(stage, flop=None):
            """profile only: close the stage that started at the previous lap.  flop: a callable, evaluated AFTER the clock was read and
            outside every stage (the FLOP counters build whole plans on the host: 0.3-0.7 s for a 68-frame batch)"""
            if prof is not None:
                torch.cuda.synchronize(dev)
                acc = prof.setdefault(stage, [0.0, 0.0])
                acc[0] += time.perf_counter() - lap.t0
                if flop is not None:
                    acc[1] += flop()
                mk = getattr(self, "stage_marker", None)
                if mk is not None:
                    mk(stage)                 # scripts/stage_stats.py: a marker kernel closes the stage in a rocprofv3 kernel trace
                    torch.cuda.synchronize(dev)
                lap.t0 = time.perf_counter()

        if prof is not None:
            torch.cuda.synchronize(dev)
            lap.t0 = time.perf_counter()
        with torch.cuda.device(dev):
            # ---- flows (:217-247): every consecutive pair in both directions, fp32; cv2.COLOR_BGR2RGB (:192) inside the stem kernel
            # Pairs are independent (instance norm and the correlation volume are per pair), so the batch goes through RAFT in equal
            # runs of at most raft_max_pairs consecutive pairs -- the reference does the same for its memory's sake (short_clip_len 2
            # at this width, :221-247).  The all-pairs correlation pyramid of ONE 1920x360 pair-direction is 0.62 GB: all 67 pairs of a
            # 68-frame batch at once made the engine's workspace 125 GB; two runs of 34 + 33 pairs need 64 GB and launch the same kernels on
            # 734 000 GEMM rows instead of 1.45 million (three runs of 23 cost 3 % of the RAFT stage: profiles/r05_sixth_call.log).
            part = max(1, self.raft_lanes) if self.PRECISIONS[self.precision][0] == "f32" else 1
            spans = raft_runs(n, self.raft_max_pairs, part)
            runs = len(spans)
            rl = 1 if prof is not None else min(part, runs)                       # (a profiled call keeps the runs and issues them on one stream)
            if runs == 1:
                gt_f, gt_b = self.fix_raft.flows(bgr, iters=self.raft_iter, bgr=True)
            elif rl == 1:
                outs = [self.fix_raft.flows(bgr[a:b], iters=self.raft_iter, bgr=True) for a, b in spans]
                gt_f, gt_b = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
            else:
                # RAFT lanes (VSR_RAFT_LANES, round 5): the runs alternate over RAFT instances on their own streams -- one run's
                # memory-bound kernels (correlation lookup, GRU gates, instance norm: a tenth of the stage) under the other's GEMMs.
                # The pairs per run shrink with the lanes, so the workspaces together stay what one lane's was.
                main = torch.cuda.current_stream(dev)
                ready = torch.cuda.Event()
                ready.record(main)
                streams = self._lane_streams(rl, dev)
                engines = [self.fix_raft] + [self._lane_raft(k) for k in range(1, rl)]
                outs = []
                for j, (a, b) in enumerate(spans):
                    st = streams[j % rl]
                    if j < rl:
                        st.wait_event(ready)
                    with torch.cuda.stream(st):
                        outs.append(engines[j % rl].flows(bgr[a:b], iters=self.raft_iter, bgr=True))
                for st in streams:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    main.wait_event(ev)
                for o in outs:
                    for t_ in o:
                        t_.record_stream(main)                                          # allocated on a lane's stream, read on the caller's
                gt_f, gt_b = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
            lap("raft", lambda: sum(self.fix_raft.flops(b - a, h, w, self.raft_iter) for a, b in spans))
            # ---- flow completion (:253-281)
            flow_length, svl = n - 1, self.sub_video_length
            if flow_length > svl:
                pf, pb = [], []
                for f in range(0, flow_length, svl):
                    s_f, e_f = max(0, f - 5), min(flow_length, f + svl + 5)
                    ps, pe = max(0, f) - s_f, e_f - min(flow_length, f + svl)
                    cf, cb = self.fix_flow_complete.complete(gt_f[s_f:e_f].contiguous(), gt_b[s_f:e_f].contiguous(), fm_dev[s_f:e_f + 1].contiguous())
                    pf.append(cf[ps:e_f - s_f - pe])
                    pb.append(cb[ps:e_f - s_f - pe])
                pred_f, pred_b = torch.cat(pf).contiguous(), torch.cat(pb).contiguous()
            else:
                pred_f, pred_b = self.fix_flow_complete.complete(gt_f, gt_b, fm_dev)
            if prof is not None:
                spans = [min(flow_length, f + svl + 5) - max(0, f - 5) + 1 for f in range(0, flow_length, svl)] if flow_length > svl else [n]
                lap("flow_completion", lambda: sum(self.fix_flow_complete.flops(t_, h, w) for t_ in spans))
            # ---- image propagation (:283-315)
            masked_frames = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev)
            check(lib.vsr_pp_prepare_frames(P(bgr), P(md1), n, h, w, P(masked_frames), stream()))
            updated_frames = torch.empty_like(masked_frames)
            sip = min(100, svl)
            if n > sip:
                um = []
                for f in range(0, n, sip):
                    s_f, e_f = max(0, f - 10), min(n, f + sip + 10)
                    ps, pe = max(0, f) - s_f, e_f - min(n, f + sip)
                    prop, upd = self.model.img_propagation(masked_frames[s_f:e_f].contiguous(), pred_f[s_f:e_f - 1].contiguous(),
                                                           pred_b[s_f:e_f - 1].contiguous(), md_dev[s_f:e_f].contiguous())
                    sub = torch.empty_like(prop)
                    check(lib.vsr_pp_compose_frames(P(bgr[s_f:e_f]), P(md1), P(prop), e_f - s_f, h, w, P(sub), stream()))
                    updated_frames[s_f + ps:e_f - pe] = sub[ps:e_f - s_f - pe]
                    um.append(upd[ps:e_f - s_f - pe])
                updated_masks = torch.cat(um).contiguous()
            else:
                prop, updated_masks = self.model.img_propagation(masked_frames, pred_f, pred_b, md_dev)
                check(lib.vsr_pp_compose_frames(P(bgr), P(md1), P(prop), n, h, w, P(updated_frames), stream()))
            lap("other")
            # ---- feature propagation + transformer over sliding neighbour windows (:317-358)
            comp = torch.empty_like(bgr)
            visited = [False] * n
            stride = self.neighbor_length // 2
            ref_num = svl // self.ref_stride if n > svl else -1
            flags_cache = {}
            # The window's prediction is blended in under the dilated mask only (vsr_pp_blend_window below, :350-357): with the promise
            # about those rows / columns the generator's decoder runs on what they depend on (vsr_pp_forward_box; whole groups of
            # eight, so that the boxes of a video that differ by a few pixels share their plans).  GPU-tested in round 5, default ON
            # (VSR_PP_DECODE_BOX=0: no promise).
            box = None
            if switches.on("VSR_PP_DECODE_BOX") and md.any():
                ys, xs = np.flatnonzero(md.any(axis=1)), np.flatnonzero(md.any(axis=0))
                box = (int(ys[0]) // 8 * 8, min(h, (int(ys[-1]) + 8) // 8 * 8), int(xs[0]) // 8 * 8, min(w, (int(xs[-1]) + 8) // 8 * 8))
            windows = []
            for f in range(0, n, stride):
                nb = [i for i in range(max(0, f - stride), min(n, f + stride + 1))]
                windows.append((nb, get_ref_index(f, nb, n, self.ref_stride, ref_num)))
            # The encoder (and a reference frame's soft split) is a per-frame function and the windows overlap: with the switch on every
            # frame is encoded once per call instead of once per window it appears in (vsr_pp_encode / vsr_pp_forward_cached).  GPU-tested in
            # round 5, default ON (VSR_PP_ENC_CACHE=0: the encoder once per window, as InpaintGenerator.forward does).
            enc_cache = None
            if switches.on("VSR_PP_ENC_CACHE"):
                calls, feat_slot, tok_slot = encoder_cache_plan(windows)
                fc, tc = [], []
                for cids, ntok in calls:
                    a, b = self.model.encode(updated_frames[cids].contiguous(), md_dev[cids].contiguous(), updated_masks[cids].contiguous(), ntok)
                    fc.append(a)
                    tc.append(b)
                enc_cache = (torch.cat(fc), torch.cat(tc) if tok_slot else None)
                if prof is not None:
                    lap("generator", lambda: sum(self.model.plan_flops(len(cids), ntok, h, w, [], None, 1) for cids, ntok in calls))
            # Generator lanes (VSR_PP_LANES, round 5): the windows are independent until their predictions are blended into `comp` in
            # window order -- as the STTN engine's window lanes, window k runs on lane k % lanes: its own generator instance (own
            # workspace, same weights) on its own stream, the blends chained by events in window order.  What overlaps is one
            # window's small / memory-bound launches (propagation steps, softmax, fold) with the other's GEMMs.
            lanes = 1 if (prof is not None or len(windows) < 2) else max(1, min(self.gen_lanes, len(windows)))
            engines = [self.model] + [self._lane_model(k) for k in range(1, lanes)]
            if lanes > 1:
                main = torch.cuda.current_stream(dev)
                ready = torch.cuda.Event()
                ready.record(main)                                                              # flows, masks, encoder cache are complete
                lane_streams = self._lane_streams(lanes, dev)
                for st in lane_streams:
                    st.wait_event(ready)
            # which frames a window sees first is a function of the window order alone
            firsts = []
            for nb, _ in windows:
                firsts.append([0 if visited[i] else 1 for i in nb])
                for i in nb:
                    visited[i] = True
            for l_t in sorted({len(nb) for nb, _ in windows}):                              # the same mask on every frame
                flags_cache[l_t] = self.model.window_flags(np.repeat(md[None], l_t, 0))
            blend_ev = [None] * len(windows)                                                # CUDA event after window k's blend
            blend_set = [threading.Event() for _ in windows]                                # ... and "it has been recorded" (host side)

            def run_window(k):
                nb, ref = windows[k]
                ids, l_t, eng = nb + ref, len(nb), engines[k % lanes]
                ctx = torch.cuda.stream(lane_streams[k % lanes]) if lanes > 1 else contextlib.nullcontext()
                try:
                    with torch.cuda.device(dev), ctx:
                        if enc_cache is not None:
                            pred = eng.forward_cached(enc_cache[0], enc_cache[1], [feat_slot[i] for i in nb] + [tok_slot[i] for i in ref],
                                                      pred_f[nb[:-1]].contiguous(), pred_b[nb[:-1]].contiguous(), md_dev[ids].contiguous(),
                                                      updated_masks[ids].contiguous(), l_t, h, w, flags_cache[l_t], box=box)
                        else:
                            pred = eng.forward(updated_frames[ids].contiguous(), pred_f[nb[:-1]].contiguous(), pred_b[nb[:-1]].contiguous(),
                                               md_dev[ids].contiguous(), updated_masks[ids].contiguous(), l_t, flags=flags_cache[l_t], box=box)
                        idx = torch.tensor(nb, dtype=torch.int32).to(dev, non_blocking=True)
                        first = torch.tensor(firsts[k], dtype=torch.int32).to(dev, non_blocking=True)
                        cur = torch.cuda.current_stream(dev)
                        if lanes > 1 and k > 0:
                            blend_set[k - 1].wait()                                         # (threaded lanes: window k - 1 runs on another thread)
                            if blend_ev[k - 1] is not None:
                                cur.wait_event(blend_ev[k - 1])                             # the running average is taken in window order
                        check(lib.vsr_pp_blend_window(P(pred), P(bgr), P(md1), P(idx), P(first), l_t, h, w, P(comp), C.c_void_p(cur.cuda_stream)))
                        if lanes > 1:
                            ev = torch.cuda.Event()
                            ev.record(cur)
                            blend_ev[k] = ev
                finally:
                    blend_set[k].set()                                                      # also after an error: nobody waits forever

            # In the guarded arithmetics (fp16 operands / split-half) every generator call ends with a read of the range flag, i.e. the
            # host waits for the window: the lanes then need a host thread each, or the second lane would never be fed while the first
            # one is waited for.  (ctypes releases the GIL for the duration of a library call.)
            threaded = lanes > 1 and self.PRECISIONS[self.precision][2] != "f32"
            try:
                if threaded:
                    from concurrent.futures import ThreadPoolExecutor

                    pools = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"vsr-pp-lane{j}") for j in range(lanes)]
                    try:
                        futs = [pools[k % lanes].submit(run_window, k) for k in range(len(windows))]
                        for f in futs:
                            f.result()
                    finally:
                        for pl in pools:
                            pl.shutdown(wait=True)
                else:
                    for k in range(len(windows)):
                        run_window(k)
                        if prof is not None:
                            nb, ref = windows[k]
                            ids, l_t = nb + ref, len(nb)
                            key = (len(ids), l_t, enc_cache is not None)

                            def window_flops(key=key, ids=ids, l_t=l_t):
                                if key not in flags_cache:
                                    flags_cache[key] = self.model.plan_flops(len(ids), l_t, h, w, flags_cache[l_t], box, 2 if enc_cache is not None else 0)
                                return flags_cache[key]

                            lap("generator", window_flops)
            finally:
                # also when a window raised: tensors allocated on the caller's stream (flows, encoder cache, comp) are read
                # and written by kernels still queued on the lane streams -- the caller's stream must end behind every lane before they
                # can go out of scope, or the allocator hands their memory to the next batch under those kernels
                if lanes > 1:
                    for st in lane_streams:
                        ev = torch.cuda.Event()
                        ev.record(st)
                        main.wait_event(ev)
            if resident:
                return comp
            out = comp.cpu().numpy()                                                        # already BGR (:360)
        return [out[i] for i in range(n)]

    def __call__(self, input_frames, input_mask):
        mask = input_mask[:, :, None]
        H_ori, W_ori = mask.shape[:2]
        split_h = int(W_ori * 3 / 16)
        inpaint_area = get_inpaint_area_by_mask(W_ori, H_ori, split_h, mask, multiple=8)
        if isinstance(input_frames, torch.Tensor):
            # the HBM-resident loop (tools/resident.py): a uint8 [n,H,W,3] device tensor, inpainted in place.  As in the list form
            # every strip is cut from the frames as they came in and written back afterwards
            comps = [self.inpaint(input_frames[:, y0:y1, x0:x1].contiguous(), mask[y0:y1, x0:x1, :]) for y0, y1, x0, x1 in inpaint_area]
            for (y0, y1, x0, x1), comp in zip(inpaint_area, comps):
                input_frames[:, y0:y1, x0:x1] = comp
            return input_frames
        frames_hr = [f.copy() for f in input_frames]
        if not inpaint_area:
            return frames_hr
        comps = {}
        for k, (y0, y1, x0, x1) in enumerate(inpaint_area):
            comps[k] = self.inpaint([f[y0:y1, x0:x1, :] for f in frames_hr], mask[y0:y1, x0:x1, :])
        for j, frame in enumerate(frames_hr):
            for k, (y0, y1, x0, x1) in enumerate(inpaint_area):
                frame[y0:y1, x0:x1, :] = comps[k][j]
        return frames_hr
