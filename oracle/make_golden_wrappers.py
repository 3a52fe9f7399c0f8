"""Generate tests/golden/wrappers.npz + wrappers.json by EXECUTING the reference's wrapper code (build container only).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_wrappers

What runs is the reference's own source (through oracle/ref_exec.py: cv2 shim = the restated cv2 primitives, a config namespace,
one log-line f-string of sttn_auto_inpaint.py rewritten for Python 3.10):

  * backend/tools/inpaint_tools.py          create_mask, get_inpaint_area_by_mask (incl. test/test.png's box and multiple=8)
  * backend/inpaint/utils/lama_util.py      get_image, pad_img_to_modulo, prepare_img_and_mask
  * backend/inpaint/lama_inpaint.py         LamaInpaint.inpaint / ._inpaint_batch / .__call__ around oracle.lama.StandInLama
  * backend/inpaint/sttn_auto_inpaint.py    STTNInpaint.__call__, STTNAutoInpaint.__call__ (two chunks, with and without A/B sections)
  * backend/inpaint/sttn_det_inpaint.py     STTNDetInpaint.__call__
with the reference's own InpaintGenerator modules and the synthetic checkpoints of vsr_amd.synth.  tests/test_oracle_wrappers.py
holds the oracle's restated wrappers (and the product's host helpers) to these outputs.
"""
import json
import os
import sys

import numpy as np
import torch

from . import ref_exec
from .lama import StandInLama

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# shared with tests/test_oracle_wrappers.py -------------------------------------------------------------------
STTN_CFG = dict(sttnNeighborStride=2, sttnReferenceLength=3, sttnMaxLoadNum=7)
AUTO_CLIP = dict(n=12, H=180, W=320, box=(140, 170, 40, 280), seed=31)          # split_h = 60
AUTO_AB = [(2, 6), (8, 11)]                                                      # range(a, b) sections
DET_CLIP = dict(n=5, H=180, W=320, box=(130, 160, 30, 290), seed=32)            # split_h = int(320*5/18) = 88
LAMA_CLIP = dict(n=9, H=182, W=330, box=(140, 170, 40, 280), seed=33)           # split_h = 61 -> padded to 64, W 330 -> 336
PP_CLIP = dict(n=7, H=288, W=704, box=(236, 268, 120, 600), seed=5)             # strip 704x132 -> x8-aligned 704x136


def area_cases():
    rng = np.random.default_rng(77)
    cases = [dict(H=480, W=852, h=int(852 * 3 / 16), boxes=[(111, 766, 373, 452)], multiple=1),    # test/test.png's box
             dict(H=480, W=852, h=int(852 * 3 / 16), boxes=[(111, 766, 373, 452)], multiple=8),
             dict(H=1080, W=1920, h=360, boxes=[(288, 1632, 950, 1070)], multiple=1),
             dict(H=1080, W=1920, h=360, boxes=[(288, 1632, 950, 1070)], multiple=8),
             dict(H=1080, W=1920, h=533, boxes=[(288, 1632, 950, 1070)], multiple=1),
             dict(H=720, W=1280, h=240, boxes=[(192, 1088, 620, 700), (300, 900, 30, 70)], multiple=1),
             dict(H=480, W=852, h=159, boxes=[(100, 760, 400, 450), (200, 600, 30, 60)], multiple=1),
             dict(H=360, W=640, h=120, boxes=[(0, 639, 0, 20)], multiple=1),
             dict(H=360, W=640, h=120, boxes=[(10, 600, 345, 359)], multiple=8),
             dict(H=360, W=640, h=120, boxes=[(10, 200, 100, 130), (300, 600, 150, 200)], multiple=1),
             dict(H=360, W=640, h=120, boxes=[(10, 200, 100, 130), (300, 600, 240, 300)], multiple=8),
             dict(H=360, W=640, h=120, boxes=[(10, 12, 100, 101)], multiple=1),
             dict(H=250, W=400, h=75, boxes=[(50, 300, 60, 200)], multiple=1),                      # island taller than the strip
             dict(H=250, W=400, h=75, boxes=[(50, 300, 60, 200)], multiple=8),
             dict(H=101, W=403, h=75, boxes=[(50, 300, 40, 80)], multiple=8)]
    for _ in range(25):
        H, W = int(rng.integers(120, 500)), int(rng.integers(200, 900))
        boxes = []
        for _ in range(int(rng.integers(1, 4))):
            x1, y1 = int(rng.integers(0, W - 20)), int(rng.integers(0, H - 10))
            boxes.append((x1, min(W + 5, x1 + int(rng.integers(3, 400))), y1, min(H + 5, y1 + int(rng.integers(2, 80)))))
        cases.append(dict(H=H, W=W, h=int(W * 3 / 16), boxes=boxes, multiple=int(rng.choice([1, 1, 8]))))
    return cases


class _Writer:
    def __init__(self):
        self.frames = []

    def write(self, f):
        self.frames.append(np.array(f, copy=True))

    def release(self):
        pass


class _Remover:
    gui_mode = False

    def __init__(self, ab):
        self.ab_sections = ab
        self.video_writer = _Writer()
        self.ticks = 0

    def update_progress(self, tbar, increment):
        self.ticks += increment


def propainter_fixture():
    """backend/inpaint/propainter_inpaint.py executed: PropainterInpaint.__call__ / .inpaint / read_mask / get_ref_index on the
    CPU (fp32 path, :146-147) with the reference's RAFT_bi, RecurrentFlowCompleteNet and InpaintGenerator modules, the synthetic
    checkpoints of vsr_amd.synth written as the three .pth files it loads, torchvision.ops.deform_conv2d from oracle/deform_conv.py
    (torchvision is absent: that operator stays unpinned), 20 RAFT iterations.  -> tests/golden/wrappers_propainter.npz"""
    import tempfile

    from oracle.deform_conv import deform_conv2d
    from vsr_amd.synth import make_clip, make_propainter_state_dict, make_raft_state_dict, make_rfc_state_dict

    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cv2, cfg = ref_exec.install()
    sys.modules["torchvision"].ops.deform_conv2d = deform_conv2d
    tools = ref_exec.load_module("backend.tools.inpaint_tools", "backend/tools/inpaint_tools.py")
    d = tempfile.mkdtemp(prefix="vsr_pp_golden_")
    torch.save({"module." + k: torch.from_numpy(v) for k, v in make_raft_state_dict(0).items()}, os.path.join(d, "raft-things.pth"))
    torch.save({k: torch.from_numpy(v) for k, v in make_rfc_state_dict(0).items()}, os.path.join(d, "recurrent_flow_completion.pth"))
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in make_propainter_state_dict(0).items()}, os.path.join(d, "ProPainter.pth"))
    pp = ref_exec.load_module("backend.inpaint.propainter_inpaint", "backend/inpaint/propainter_inpaint.py")
    plug = pp.PropainterInpaint(torch.device("cpu"), d, sub_video_length=70)            # main.py:171
    assert plug.use_half is False and plug.raft_iter == 20
    c = PP_CLIP
    clip = make_clip(c["n"], c["H"], c["W"], c["box"], seed=c["seed"])
    b = c["box"]
    mask = tools.create_mask((c["H"], c["W"]), [(b[2], b[3], b[0], b[1])])
    out = np.stack(plug([f.copy() for f in clip], mask))
    changed = (out != clip).any(axis=(0, 3))
    ys, xs = np.nonzero(changed)
    bbox = [int(ys.min()), int(ys.max()) + 1, int(xs.min()), int(xs.max()) + 1]
    areas = tools.get_inpaint_area_by_mask(c["W"], c["H"], int(c["W"] * 3 / 16), mask[:, :, None], multiple=8)
    print("propainter: areas", areas, "changed bbox", bbox, "changed fraction", float(changed.mean()))
    np.savez_compressed(os.path.join(OUT, "wrappers_propainter.npz"), bbox=np.array(bbox), area=np.array(areas[0]),
                        out=out[:, bbox[0]:bbox[1], bbox[2]:bbox[3]], changed=np.packbits(changed))


def main():
    if "--propainter" in sys.argv:
        return propainter_fixture()
    from vsr_amd.synth import make_clip, make_state_dict

    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cv2, cfg = ref_exec.install(**STTN_CFG)
    tools = ref_exec.load_module("backend.tools.inpaint_tools", "backend/tools/inpaint_tools.py")
    js, npz = {"areas": [], "create_mask": []}, {}

    # ---- inpaint_tools.create_mask / get_inpaint_area_by_mask -------------------------------------------------
    for c in area_cases():
        mask = tools.create_mask((c["H"], c["W"]), [tuple(b) for b in c["boxes"]])
        areas = tools.get_inpaint_area_by_mask(c["W"], c["H"], c["h"], mask[:, :, None], multiple=c["multiple"])
        js["areas"].append(dict(c, out=[list(int(v) for v in a) for a in areas]))
        ys, xs = np.nonzero(mask)
        js["create_mask"].append(dict(H=c["H"], W=c["W"], boxes=c["boxes"], set=int(mask.astype(bool).sum()),
                                      bbox=[int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())] if ys.size else None,
                                      rowsum=np.packbits(mask.any(axis=1)).tolist(), values=sorted(int(v) for v in np.unique(mask))))

    # ---- lama_util ------------------------------------------------------------------------------------------------
    lu = ref_exec.load_module("backend.inpaint.utils.lama_util", "backend/inpaint/utils/lama_util.py")
    rng = np.random.default_rng(91)
    img = rng.integers(0, 256, (13, 21, 3), dtype=np.uint8)
    msk = (rng.random((13, 21, 1)) > 0.6).astype(np.uint8) * 255
    npz["lu_img"], npz["lu_mask"] = img, msk
    npz["lu_get_image"] = lu.get_image(img)
    npz["lu_pad8"] = lu.pad_img_to_modulo(lu.get_image(img), 8)
    pi, pm = lu.prepare_img_and_mask(img, msk, torch.device("cpu"))
    npz["lu_prep_img"], npz["lu_prep_mask"] = pi.numpy(), pm.numpy()
    js["lu_prep_mask_dtype"] = str(pm.dtype)

    # ---- LamaInpaint around the stand-in module ----------------------------------------------------------------
    torch.jit.load = lambda path, map_location=None: StandInLama(5)         # lama_inpaint.py:13
    lm = ref_exec.load_module("backend.inpaint.lama_inpaint", "backend/inpaint/lama_inpaint.py")
    plug = lm.LamaInpaint(torch.device("cpu"), "stand-in")
    c = LAMA_CLIP
    clip = make_clip(c["n"], c["H"], c["W"], c["box"], seed=c["seed"])
    b = c["box"]
    mask = tools.create_mask((c["H"], c["W"]), [(b[2], b[3], b[0], b[1])])
    out = np.stack(plug([f for f in clip], mask))
    areas = tools.get_inpaint_area_by_mask(c["W"], c["H"], int(c["W"] * 3 / 16), mask[:, :, None])
    (y0, y1, _, _), = areas
    assert np.array_equal(out[:, :y0], clip[:, :y0]) and np.array_equal(out[:, y1:], clip[:, y1:])
    npz["lama_call_strip"] = out[:, y0:y1]
    js["lama_area"] = [int(y0), int(y1)]
    npz["lama_single"] = plug.inpaint(clip[0], mask)                        # main.py:220,233,364: whole frame, no strip
    npz["lama_batch1"] = plug._inpaint_batch([clip[1][y0:y1]], [mask[y0:y1, :, None]])[0]

    # ---- sttn-auto wrappers ------------------------------------------------------------------------------------
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(0, "auto").items()}
    torch.save({"netG": sd}, "/tmp/vsr_golden_auto.pth")
    ref_exec.load_module("backend.tools.video_io", "backend/tools/video_io.py")
    sa = ref_exec.load_module("backend.inpaint.sttn_auto_inpaint", "backend/inpaint/sttn_auto_inpaint.py", ref_exec.STTN_AUTO_PATCH)
    c = AUTO_CLIP
    clip = make_clip(c["n"], c["H"], c["W"], c["box"], seed=c["seed"])
    b = c["box"]
    mask = tools.create_mask((c["H"], c["W"]), [(b[2], b[3], b[0], b[1])])
    ref_exec.VIDEOS["/tmp/vsr_golden_clip.mp4"] = clip
    areas = tools.get_inpaint_area_by_mask(c["W"], c["H"], int(c["W"] * 3 / 16), cv2.threshold(mask, 127, 1, 0)[1][:, :, None])
    (y0, y1, _, _), = areas
    js["auto_area"] = [int(y0), int(y1)]
    for tag, ab in (("all", None), ("ab", [range(a, e) for a, e in AUTO_AB])):
        plug = sa.STTNAutoInpaint(torch.device("cpu"), "/tmp/vsr_golden_auto.pth", "/tmp/vsr_golden_clip.mp4")
        assert plug.clip_gap == 7
        rem = _Remover(ab)
        plug(input_mask=mask, input_sub_remover=rem, tbar=object())
        out = np.stack(rem.video_writer.frames)
        assert out.shape == clip.shape and rem.ticks == c["n"], (out.shape, rem.ticks)
        assert np.array_equal(out[:, :y0], clip[:, :y0]) and np.array_equal(out[:, y1:], clip[:, y1:])
        npz[f"auto_call_{tag}_strip"] = out[:, y0:y1]
        print("sttn-auto __call__", tag, "changed pixels:", int((out != clip).sum()))
    gen = sa.STTNInpaint(torch.device("cpu"), "/tmp/vsr_golden_auto.pth")
    out = np.stack(gen([f for f in clip[:6]], mask))
    assert np.array_equal(out[:, :y0], clip[:6, :y0])
    npz["auto_plugin_strip"] = out[:, y0:y1]
    comps = gen.inpaint([cv2.resize(f[y0:y1], (640, 120)) for f in clip[:6]])                   # :122-164 on its own
    js["auto_inpaint_dtypes"] = [str(x.dtype) for x in comps]
    q = np.stack([comps[i].astype(np.float32) for i in (0, 2, 5)]) * 4   # frames 0, 2, 5; averages of uint8 pairs of pairs: exact multiples of 1/4
    assert np.array_equal(q, np.rint(q)) and q.max() < 65536
    npz["auto_inpaint_x4"] = q.astype(np.uint16)

    # ---- sttn-det wrapper --------------------------------------------------------------------------------------
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(1, "det").items()}
    torch.save({"netG": sd}, "/tmp/vsr_golden_det.pth")
    sdm = ref_exec.load_module("backend.inpaint.sttn_det_inpaint", "backend/inpaint/sttn_det_inpaint.py")
    c = DET_CLIP
    clip = make_clip(c["n"], c["H"], c["W"], c["box"], seed=c["seed"])
    b = c["box"]
    mask = tools.create_mask((c["H"], c["W"]), [(b[2], b[3], b[0], b[1])])
    plug = sdm.STTNDetInpaint(torch.device("cpu"), "/tmp/vsr_golden_det.pth")
    frames_in = [f.copy() for f in clip]
    out = np.stack(plug(frames_in, mask))
    areas = tools.get_inpaint_area_by_mask(c["W"], c["H"], int(c["W"] * 5 / 18), mask[:, :, None])
    (y0, y1, _, _), = areas
    assert np.array_equal(out[:, :y0], clip[:, :y0]) and np.array_equal(out[:, y1:], clip[:, y1:])
    js["det_area"] = [int(y0), int(y1)]
    js["det_inputs_mutated"] = bool(any(not np.array_equal(a, b2) for a, b2 in zip(frames_in, clip)))
    npz["det_call_strip"] = out[:, y0:y1]

    np.savez_compressed(os.path.join(OUT, "wrappers.npz"), **npz)
    with open(os.path.join(OUT, "wrappers.json"), "w") as f:
        json.dump(js, f)
    print("wrote wrappers.npz", os.path.getsize(os.path.join(OUT, "wrappers.npz")) // 1024, "KiB")


if __name__ == "__main__":
    sys.exit(main())
