"""Generate tests/golden/* from the REFERENCE implementation (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

Imports the reference's own nn.Modules from /root/reference (read-only; import stubs for cv2 /
torchvision only satisfy unrelated top-level imports: backend/inpaint/utils/__init__.py star-imports
a cv2-using helper, network_sttn.py:8 imports unused torchvision.models), loads the synthetic
weights of oracle/weights.py with load_state_dict(strict=True) -- which also pins the key names
and shapes -- and records outputs on seeded inputs.  /root/reference does not exist on the GPU
box: tests only read the committed fixtures.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class _Permissive(types.ModuleType):
    """Stub module: any attribute is another stub, any call returns a stub (for unrelated top-level imports)."""

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        m = _Permissive(self.__name__ + "." + n)
        setattr(self, n, m)
        return m

    def __call__(self, *a, **k):
        return _Permissive("call")


class _Item:
    def __init__(self, v):
        self.value = v


def _import_reference():
    for n in ("cv2", "torchvision", "torchvision.models", "onnxruntime", "qfluentwidgets", "paddleocr", "fsplit",
              "fsplit.filesplit"):
        sys.modules.setdefault(n, _Permissive(n))
    cfg = _Permissive("backend.config")               # qfluentwidgets-free stand-in (defaults of backend/config.py:59-68)
    cfg.config = types.SimpleNamespace(subtitleAreaPixelToleranceXPixel=_Item(20), subtitleAreaPixelToleranceYPixel=_Item(20))
    cfg.tr = {}
    cfg.BASE_DIR = "/tmp"
    sys.path.insert(0, REF)
    from backend.inpaint.sttn import auto_sttn, network_sttn
    sys.modules["backend.config"] = cfg
    sys.modules["backend.scenedetect"] = _Permissive("backend.scenedetect")
    sys.modules["backend.scenedetect.detectors"] = _Permissive("backend.scenedetect.detectors")
    from backend.tools import inpaint_tools, ocr, subtitle_detect
    inpaint_tools.subtitle_detect = subtitle_detect
    inpaint_tools.ocr = ocr
    return auto_sttn, network_sttn, inpaint_tools


def _bookkeeping_fixture(inpaint_tools):
    """Temporal bookkeeping of the detector modes, executed from the reference."""
    sd = inpaint_tools.subtitle_detect.SubtitleDetect
    rng = np.random.default_rng(2024)
    out = {"filter_and_merge": [], "expand": [], "coords": [], "unify": [], "ranges": [], "ranges_same_mask": []}
    for _ in range(60):
        n = int(rng.integers(1, 9))
        starts = np.sort(rng.choice(np.arange(1, 400), size=n, replace=False))
        iv = []
        last = 0
        for s0 in starts:
            s0 = max(int(s0), last + 1)
            e0 = s0 + int(rng.choice([0, 0, 1, 3, 8, 15, 40]))
            iv.append((s0, e0))
            last = e0
        tl = int(rng.choice([5, 10, 10, 12]))
        out["filter_and_merge"].append({"in": iv, "target": tl, "out": sd.filter_and_merge_intervals(list(iv), tl)})
        b, f = int(rng.integers(0, 6)), int(rng.integers(0, 6))
        out["expand"].append({"in": iv, "b": b, "f": f, "out": inpaint_tools.expand_frame_ranges(list(iv), b, f)})
    for _ in range(30):
        k = int(rng.integers(1, 4))
        polys = []
        for _ in range(k):
            x1, y1 = int(rng.integers(0, 500)), int(rng.integers(0, 300))
            w, h = int(rng.integers(5, 400)), int(rng.integers(5, 80))
            j = lambda: int(rng.integers(-3, 4))
            polys.append([[x1 + j(), y1 + j()], [x1 + w + j(), y1 + j()], [x1 + w + j(), y1 + h + j()], [x1 + j(), y1 + h + j()]])
        out["coords"].append({"in": polys, "out": inpaint_tools.ocr.get_coordinates([list(p) for p in polys])})
    det = sd.__new__(sd)
    for _ in range(30):
        frames = sorted(set(int(v) for v in rng.integers(1, 60, size=int(rng.integers(1, 25)))))
        base = [(100, 700, 400, 450), (120, 600, 300, 340)]
        regs = {}
        for fno in frames:
            boxes = []
            for bx in base[: int(rng.integers(1, 3))]:
                d = [int(v) for v in rng.integers(-30, 31, size=4)]
                boxes.append((bx[0] + d[0], bx[1] + d[1], bx[2] + d[2], bx[3] + d[3]))
            regs[fno] = boxes
        uni = det.unify_regions({k: list(v) for k, v in regs.items()})
        out["unify"].append({"in": {str(k): v for k, v in regs.items()}, "out": {str(k): v for k, v in uni.items()}})
        out["ranges"].append({"in": frames, "out": sd.find_continuous_ranges({k: 1 for k in frames})})
        out["ranges_same_mask"].append({"in": {str(k): v for k, v in uni.items()},
                                        "out": sd.find_continuous_ranges_with_same_mask(uni)})
    return out


def _sample(t, n=4096, seed=123):
    flat = t.detach().reshape(-1).numpy()
    idx = np.random.default_rng(seed).integers(0, flat.size, size=n)
    return idx.astype(np.int64), flat[idx].astype(np.float32)


def _raft_fixture():
    """RAFT (raft/raft.py:87-146, the "things" configuration of flow_comp_raft.py:10-24) run from the reference on
    three synthetic frames: consecutive pairs in both directions as RAFT_bi.forward does (:39-55), 20 iterations."""
    import argparse

    from vsr_amd.synth import make_flow_frames, make_raft_state_dict

    for n in ("torchvision.ops", "torchvision.transforms"):
        sys.modules.setdefault(n, _Permissive(n))
    from backend.inpaint.video.raft.raft import RAFT

    net = RAFT(argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)).eval()
    sd = make_raft_state_dict(0)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    assert sum(p.numel() for p in net.parameters()) == 5257536
    frames = make_flow_frames(3, 128, 192, seed=1)
    x = torch.from_numpy(frames).permute(0, 3, 1, 2).float().div(255) * 2 - 1   # to_tensors()(frames) * 2 - 1
    res = {}
    with torch.no_grad():
        for iters in (1, 20):
            lo_f, up_f = net(x[:-1], x[1:], iters=iters, test_mode=True)
            lo_b, up_b = net(x[1:], x[:-1], iters=iters, test_mode=True)
            res[f"low_f_{iters}"], res[f"up_f_{iters}"] = lo_f.numpy(), up_f[..., ::2, ::3].numpy()
            res[f"low_b_{iters}"], res[f"up_b_{iters}"] = lo_b.numpy(), up_b[..., ::2, ::3].numpy()
        fmap = net.fnet(x[:1])
        cmap = net.cnet(x[:1])
    np.savez_compressed(os.path.join(OUT, "raft.npz"), frames_seed=1, fmap_sub=fmap[:, ::8].numpy(), cmap_sub=cmap[:, ::8].numpy(),
                        **res)


def rfc_inputs(seed, t, h, w):
    """Seeded (flows_f, flows_b [t-1,2,h,w], masks [t,1,h,w] in {0,1}) for the flow-completion fixtures and tests."""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((2, 2, h // 8 + 2, w // 8 + 2)).astype(np.float32) * 4
    up = torch.nn.functional.interpolate(torch.from_numpy(base), size=(h, w), mode="bilinear", align_corners=True).numpy()
    ff = np.stack([up[0] + 0.3 * i + rng.standard_normal((2, h, w)).astype(np.float32) * 0.2 for i in range(t - 1)])
    fb = np.stack([up[1] - 0.2 * i + rng.standard_normal((2, h, w)).astype(np.float32) * 0.2 for i in range(t - 1)])
    masks = np.zeros((t, 1, h, w), dtype=np.float32)
    masks[:, :, h // 2: h // 2 + h // 4, w // 8: w - w // 8] = 1
    return ff.astype(np.float32), fb.astype(np.float32), masks


def _rfc_fixture():
    """RecurrentFlowCompleteNet.forward_bidirect_flow + combine_flow (recurrent_flow_completion.py:313-348) run from the
    reference with torchvision.ops.deform_conv2d provided by oracle/deform_conv.py (torchvision is absent here)."""
    from oracle.deform_conv import deform_conv2d
    from vsr_amd.synth import make_rfc_state_dict

    sys.modules["torchvision"].ops.deform_conv2d = deform_conv2d
    from backend.inpaint.video.model.recurrent_flow_completion import RecurrentFlowCompleteNet

    net = RecurrentFlowCompleteNet().eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in make_rfc_state_dict(0).items()}, strict=True)
    assert sum(p.numel() for p in net.parameters()) == 5079555
    ff, fb, masks = rfc_inputs(21, 5, 64, 96)
    tf, tb, tm = torch.from_numpy(ff)[None], torch.from_numpy(fb)[None], torch.from_numpy(masks)[None]
    with torch.no_grad():
        (pf, pb), _ = net.forward_bidirect_flow([tf, tb], tm)
        cf, cb = net.combine_flow([tf, tb], [pf, pb], tm)
    np.savez_compressed(os.path.join(OUT, "rfc.npz"), seed=21, pred_f=pf[0].numpy(), pred_b=pb[0].numpy(),
                        comb_f=cf[0, :, :, ::2, ::2].numpy(), comb_b=cb[0, :, :, ::2, ::2].numpy())


def propainter_inputs(seed, t, lt, h, w):
    """Seeded inputs of the ProPainter generator: frames [t,3,h,w] in [-1,1], masks [t,1,h,w] in {0,1}, completed flows
    [lt-1,2,h,w] (smooth, a few pixels) for the lt local frames."""
    rng = np.random.default_rng(seed)
    base = torch.from_numpy(rng.uniform(-1, 1, (t, 3, h // 4 + 1, w // 4 + 1)).astype(np.float32))
    frames = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=True).numpy()
    frames = np.clip(frames + rng.normal(0, 0.05, frames.shape).astype(np.float32), -1, 1).astype(np.float32)
    masks = np.zeros((t, 1, h, w), dtype=np.float32)
    masks[:, :, h // 2 + 2: h - h // 8, w // 12: w - w // 12] = 1
    fl = torch.from_numpy(rng.standard_normal((2, 2, h // 8 + 2, w // 8 + 2)).astype(np.float32) * 2.5)
    up = torch.nn.functional.interpolate(fl, size=(h, w), mode="bilinear", align_corners=True).numpy()
    ff = np.stack([up[0] + 0.2 * i for i in range(lt - 1)]).astype(np.float32)
    fb = np.stack([-up[0] - 0.2 * i + 0.3 * up[1] for i in range(lt - 1)]).astype(np.float32)
    return frames, masks, ff, fb


def _propainter_fixture():
    """InpaintGenerator.img_propagation + forward (propainter.py:316-378) as PropainterInpaint.inpaint chains them
    (propainter_inpaint.py:283-341), run from the reference with oracle/deform_conv.py for torchvision.ops.deform_conv2d."""
    from oracle.deform_conv import deform_conv2d
    from vsr_amd.synth import make_propainter_state_dict

    sys.modules["torchvision"].ops.deform_conv2d = deform_conv2d
    from backend.inpaint.video.model.propainter import InpaintGenerator

    net = InpaintGenerator(init_weights=False).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in make_propainter_state_dict(0).items()}, strict=True)
    assert sum(p.numel() for p in net.parameters()) == 39429667
    t, lt, h, w = 7, 5, 64, 96
    frames, masks, ff, fb = propainter_inputs(41, t, lt, h, w)
    fr, mk, tf, tb = (torch.from_numpy(a)[None] for a in (frames, masks, ff, fb))
    with torch.no_grad():
        masked = fr * (1 - mk)
        prop, upd = net.img_propagation(masked[:, :lt], (tf, tb), mk[:, :lt].clone(), "nearest")
        upd_frames = fr[:, :lt] * (1 - mk[:, :lt]) + prop.view(1, lt, 3, h, w) * mk[:, :lt]
        sel = torch.cat([upd_frames, masked[:, lt:]], 1)
        sel_upd = torch.cat([upd.view(1, lt, 1, h, w), mk[:, lt:]], 1)
        out = net(sel, (tf, tb), mk, sel_upd, lt)
    np.savez_compressed(os.path.join(OUT, "propainter.npz"), seed=41, prop=prop.view(lt, 3, h, w)[:, :, ::2, ::2].numpy(),
                        upd_mask=upd.view(lt, h, w).numpy().astype(np.uint8), out=out[0].numpy())


def _detector_graph_fixtures():
    """The PP-OCRv5 detection programs (backend/models/V5/{ch_det_fast,ch_det}/inference.json) condensed by
    vsr_amd.backend.tools.paddle_graph (op list + parameter shapes only): the GPU box has no reference mount."""
    from vsr_amd.backend.tools.paddle_graph import load_graph

    for name, out in (("ch_det_fast", "ppocr_det_fast_graph.json"), ("ch_det", "ppocr_det_graph.json")):
        g = load_graph(os.path.join(REF, "backend", "models", "V5", name, "inference.json"))
        with open(os.path.join(OUT, out), "w") as f:
            json.dump(g.to_json(), f, separators=(",", ":"))


def scene_clip(seed=5, n=90, H=120, W=780):
    """synthetic clip with hard cuts (some closer than min_scene_len to the previous one) and slow drift in between"""
    rng = np.random.default_rng(seed)
    cuts_at = {18, 25, 33, 50, 51, 70, 88}
    yy, xx = np.mgrid[0:H, 0:W]
    frames = []
    base = None
    for i in range(n):
        if base is None or i in cuts_at:
            base = rng.integers(0, 256, (H // 12 + 1, W // 12 + 1, 3)).astype(np.float64)
            shade = rng.uniform(0.3, 1.0)
        img = base[yy // 12, xx // 12] * shade + 6.0 * np.sin(0.3 * i + xx / 40.0)[..., None]
        frames.append(np.clip(img, 0, 255).astype(np.uint8))
    return np.stack(frames)


def _scene_cut_fixture():
    """backend/scenedetect's own SceneManager + ContentDetector (scene_manager.py, detectors/content_detector.py) and
    SubtitleDetect.get_scene_div_frame_no's conversion (subtitle_detect.py:158-170) run over an in-memory stream; the two cv2
    calls on the way (resize, cvtColor) are the restatements of oracle/cv2_restate.py and oracle/scene_cuts.py."""
    import importlib.util

    from . import cv2_restate, scene_cuts

    cv2 = sys.modules["cv2"]
    saved = {k: cv2.__dict__.get(k) for k in ("cvtColor", "split", "resize", "COLOR_BGR2HSV", "INTER_LINEAR")}
    cv2.COLOR_BGR2HSV, cv2.INTER_LINEAR = 40, 1
    cv2.cvtColor = lambda img, code: scene_cuts.bgr2hsv_u8(img)
    cv2.split = lambda img: [img[..., k] for k in range(img.shape[-1])]

    def _resize(img, dsize, interpolation=None):
        assert interpolation == 1
        return cv2_restate.resize_linear(img, dsize)

    cv2.resize = _resize
    sys.modules.setdefault("tqdm", _Permissive("tqdm"))

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, "backend", "scenedetect", rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    keep = {k: sys.modules.get(k) for k in ("backend.scenedetect", "backend.scenedetect.detectors")}
    pkg = types.ModuleType("backend.scenedetect")
    pkg.__path__ = []
    sys.modules["backend.scenedetect"] = pkg
    tp = types.ModuleType("backend.scenedetect._thirdparty")
    tp.__path__ = []
    sys.modules["backend.scenedetect._thirdparty"] = tp
    for mod in ("platform", "frame_timecode", "_thirdparty.simpletable", "video_stream", "stats_manager", "scene_detector", "scene_manager"):
        load("backend.scenedetect." + mod, mod.replace(".", "/") + ".py")
    cd = load("backend.scenedetect.detectors.content_detector", "detectors/content_detector.py")
    sm = sys.modules["backend.scenedetect.scene_manager"]
    FrameTimecode = sys.modules["backend.scenedetect.frame_timecode"].FrameTimecode

    class ArrayStream:                      # position semantics of VideoStreamCv2 (backends/opencv.py:189-217)
        def __init__(self, frames):
            self.frames, self.n = frames, 0
            self.frame_rate = 25.0
            self.base_timecode = FrameTimecode(0, 25.0)
            self.frame_size = (frames.shape[2], frames.shape[1])
            self.duration = self.base_timecode + frames.shape[0]

        @property
        def frame_number(self):
            return self.n

        @property
        def position(self):
            return self.base_timecode if self.n < 1 else self.base_timecode + (self.n - 1)

        def read(self, decode=True, advance=True):
            if self.n >= self.frames.shape[0]:
                return False
            f = self.frames[self.n]
            self.n += 1
            return f if decode else True

    out = {}
    for tag, kw in (("a", dict(seed=5, n=90, H=120, W=780)), ("b", dict(seed=6, n=40, H=90, W=200)), ("c", dict(seed=7, n=60, H=270, W=1030))):
        clip = scene_clip(**kw)
        det = cd.ContentDetector()
        scores = []
        orig = det.process_frame

        def spy(frame_num, frame_img, _orig=orig, _det=det, _scores=scores):
            r = _orig(frame_num, frame_img)
            _scores.append(float(_det._frame_score))
            return r

        det.process_frame = spy
        mgr = sm.SceneManager(None)
        mgr.add_detector(det)
        mgr.detect_scenes(video=ArrayStream(clip), show_progress=False)
        div = [s.frame_num + 1 for s, _ in mgr.get_scene_list(start_in_scene=False) if s.frame_num != 0]     # subtitle_detect.py:163-169
        out[tag] = {"clip": kw, "factor": int(sm.compute_downscale_factor(clip.shape[2])), "scores": scores, "div": div}
        print("scene cuts", tag, out[tag]["factor"], div)
    with open(os.path.join(OUT, "scene_cuts.json"), "w") as f:
        json.dump(out, f)
    for k, v in saved.items():
        if v is not None:
            setattr(cv2, k, v)
    for k, v in keep.items():
        if v is not None:
            sys.modules[k] = v


def main():
    from vsr_amd.synth import make_state_dict

    os.makedirs(OUT, exist_ok=True)
    auto_sttn, network_sttn, inpaint_tools = _import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))

    # ---- sttn-auto generator: encoder -> infer -> decoder -> tanh on 3 frames of 120x640 ----
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(0, "auto").items()}
    net = auto_sttn.InpaintGenerator(init_weights=False).eval()
    net.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(7)
    frames = rng.integers(0, 256, size=(3, 120, 640, 3), dtype=np.uint8)      # BGR
    x = torch.from_numpy(np.ascontiguousarray(frames[..., ::-1])).permute(0, 3, 1, 2).float().div(255) * 2 - 1
    with torch.no_grad():
        feat = net.encoder(x)
        pred = net.infer(feat)
        out = torch.tanh(net.decoder(pred[:2]))
    fi, fv = _sample(feat)
    pi, pv = _sample(pred)
    np.savez_compressed(
        os.path.join(OUT, "sttn_auto_net.npz"), frames_seed=7,
        feat_idx=fi, feat_val=fv, feat_sum=np.float64(feat.double().sum()), feat_sq=np.float64((feat.double() ** 2).sum()),
        pred_idx=pi, pred_val=pv, pred_sum=np.float64(pred.double().sum()), pred_sq=np.float64((pred.double() ** 2).sum()),
        out_sub=out[:, :, ::2, ::4].numpy().astype(np.float32),
        out_sum=np.float64(out.double().sum()), out_sq=np.float64((out.double() ** 2).sum()))

    # ---- sttn-det generator (mask input must have no effect: network_sttn.py:149) ----
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(1, "det").items()}
    net = network_sttn.InpaintGenerator(init_weights=False).eval()
    net.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(8)
    x = torch.from_numpy(rng.random((2, 3, 240, 432), dtype=np.float32) * 2 - 1)
    masks = torch.from_numpy((rng.random((2, 1, 240, 432)) > 0.7).astype(np.float32))
    with torch.no_grad():
        feat = net.encoder(x)
        pred = net.infer(feat, masks)
        out = torch.tanh(net.decoder(pred[:1]))
    pi, pv = _sample(pred)
    np.savez_compressed(
        os.path.join(OUT, "sttn_det_net.npz"), x_seed=8,
        pred_idx=pi, pred_val=pv, pred_sum=np.float64(pred.double().sum()), pred_sq=np.float64((pred.double() ** 2).sum()),
        out_sub=out[:, :, ::4, ::4].numpy().astype(np.float32),
        out_sum=np.float64(out.double().sum()), out_sq=np.float64((out.double() ** 2).sum()))

    _raft_fixture()
    _rfc_fixture()
    _propainter_fixture()
    _detector_graph_fixtures()
    _scene_cut_fixture()

    # ---- batch_generator (tools/inpaint_tools.py:7-29), executed from the reference ----
    cases = [(1200, 50), (300, 50), (600, 50), (1200, 70), (49, 50), (50, 50), (51, 50), (75, 50), (1, 50), (0, 50),
             (10, 3), (7, 1), (99, 10), (100, 10), (101, 10)]
    res = {f"{n},{m}": [len(b) for b in inpaint_tools.batch_generator(list(range(n)), m)] for n, m in cases}
    with open(os.path.join(OUT, "batch_generator.json"), "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)
    with open(os.path.join(OUT, "bookkeeping.json"), "w") as f:
        json.dump(_bookkeeping_fixture(inpaint_tools), f)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
