"""Run the REFERENCE's own wrapper code in the build container (fixture generation only; test infrastructure).

The reference's plugin wrappers (backend/inpaint/{sttn_auto,sttn_det,lama,propainter}_inpaint.py, backend/tools/inpaint_tools.py,
backend/inpaint/utils/lama_util.py) import cv2, torchvision, qfluentwidgets (through backend.config) and -- sttn_auto_inpaint.py:245
-- use a Python >= 3.12 f-string; none of that exists in this image.  This module makes them executable anyway:

  * ``cv2``: a shim module whose functions are the numpy restatements of oracle/cv2_restate.py (resize, threshold, rectangle,
    connectedComponentsWithStats) plus channel swaps and an array-backed VideoCapture.  The cv2 primitives therefore stay
    PARITY UNPINNED (opencv-python is absent); everything AROUND them -- loops, window schedule, truncations, averaging order,
    blending, padding, batching -- is the reference's code, executed.
  * ``torchvision.transforms.Compose``: three lines.
  * ``backend.config``: a plain namespace with the defaults of backend/config.py.
  * sttn_auto_inpaint.py is compiled from its source text with the one nested-quote f-string of line 245 (a log line) rewritten.

/root/reference does not exist on the GPU box; only oracle/make_golden*.py import this module.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np

from . import cv2_restate as cv2r

REF = "/root/reference"


class _Permissive(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        m = _Permissive(self.__name__ + "." + n)
        setattr(self, n, m)
        return m

    def __call__(self, *a, **k):
        return _Permissive("call")


class Item:
    def __init__(self, v):
        self.value = v


VIDEOS = {}          # path -> uint8 [N,H,W,3] (what the VideoCapture shim "decodes")


class _VideoCapture:
    def __init__(self, path):
        self.frames = VIDEOS[path]
        self.pos = 0

    def get(self, prop):
        n, h, w, _ = self.frames.shape
        return {3: float(w), 4: float(h), 5: 25.0, 7: float(n)}[prop]

    def isOpened(self):
        return True

    def read(self):
        if self.pos >= len(self.frames):
            return False, None
        f = self.frames[self.pos].copy()
        self.pos += 1
        return True, f

    def release(self):
        pass


def make_cv2():
    cv2 = types.ModuleType("cv2")
    cv2.THRESH_BINARY, cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.INTER_AREA = 0, 0, 1, 3
    cv2.COLOR_BGR2RGB, cv2.COLOR_RGB2BGR, cv2.COLOR_BGR2GRAY = 4, 4, 6
    cv2.CC_STAT_LEFT, cv2.CC_STAT_TOP, cv2.CC_STAT_WIDTH, cv2.CC_STAT_HEIGHT, cv2.CC_STAT_AREA = 0, 1, 2, 3, 4
    cv2.CAP_PROP_FRAME_WIDTH, cv2.CAP_PROP_FRAME_HEIGHT, cv2.CAP_PROP_FPS, cv2.CAP_PROP_FRAME_COUNT = 3, 4, 5, 7

    def threshold(src, thresh, maxval, typ):
        assert typ == 0
        return float(thresh), cv2r.threshold_binary(src, thresh, maxval)

    def resize(img, dsize, interpolation=1, **kw):
        assert interpolation == 1 and dsize is not None
        a = img if img.ndim == 3 else img[:, :, None]
        out = cv2r.resize_linear(np.ascontiguousarray(a), tuple(int(v) for v in dsize))
        return out[:, :, 0] if out.shape[2] == 1 else out            # cv2 drops a single channel axis

    def cvtColor(img, code):
        assert code == 4, "only the 3-channel swap is on the path"
        return np.ascontiguousarray(img[:, :, ::-1])

    def rectangle(mask, pt1, pt2, color, thickness=1):
        assert thickness == -1
        return cv2r.rectangle_filled(mask, pt1, pt2, color[0] if isinstance(color, (tuple, list)) else color)

    def connectedComponentsWithStats(binary, connectivity=8):
        return cv2r.connected_components_with_stats(binary[:, :, 0] if binary.ndim == 3 else binary, connectivity)   # cv2 takes HxWx1

    cv2.threshold, cv2.resize, cv2.cvtColor, cv2.rectangle = threshold, resize, cvtColor, rectangle
    cv2.connectedComponentsWithStats = connectedComponentsWithStats
    cv2.VideoCapture = _VideoCapture
    cv2.setNumThreads = lambda n: None
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
    return cv2


def make_config(**over):
    """backend/config.py defaults the wrappers read (:55-103); `over` replaces values."""
    vals = dict(sttnNeighborStride=5, sttnReferenceLength=10, sttnMaxLoadNum=50, propainterMaxLoadNum=70,
                subtitleAreaDeviationPixel=10, subtitleYXAxisDifferencePixel=10, subtitleAreaPixelToleranceXPixel=20,
                subtitleAreaPixelToleranceYPixel=20, subtitleTimelineBackwardFrameCount=3, subtitleTimelineForwardFrameCount=3)
    vals.update(over)
    cfg = types.SimpleNamespace(**{k: Item(v) for k, v in vals.items()})
    cfg.getSttnMaxLoadNum = lambda: max(cfg.sttnMaxLoadNum.value, cfg.sttnNeighborStride.value * cfg.sttnReferenceLength.value)   # config.py:89-94
    return cfg


def install(**config_over):
    """Put the shims into sys.modules and return (cv2 shim, config namespace)."""
    cv2 = make_cv2()
    sys.modules["cv2"] = cv2
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    tr.Compose = Compose
    tv.transforms = tr
    tv.models = _Permissive("torchvision.models")
    tv.ops = _Permissive("torchvision.ops")
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.models": tv.models, "torchvision.ops": tv.ops})
    for n in ("onnxruntime", "qfluentwidgets", "paddleocr", "fsplit", "fsplit.filesplit"):
        sys.modules.setdefault(n, _Permissive(n))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import backend  # noqa: F401  (namespace of the reference)

    cfgmod = types.ModuleType("backend.config")
    cfgmod.config = make_config(**config_over)
    cfgmod.tr = {}
    cfgmod.BASE_DIR = "/tmp"
    sys.modules["backend.config"] = cfgmod
    backend.config = cfgmod
    hw = types.ModuleType("backend.tools.hardware_accelerator")

    class HardwareAccelerator:                       # tools/hardware_accelerator.py: only the free-VRAM probe is read on this path
        @classmethod
        def instance(cls):
            return cls()

        def get_available_vram_mb(self):
            return 0                                 # "unknown" -> the clip_gap clamp of sttn_auto_inpaint.py:228-238 is skipped

    hw.HardwareAccelerator = HardwareAccelerator
    sys.modules["backend.tools.hardware_accelerator"] = hw
    return cv2, cfgmod.config


def load_module(name, rel, patch=None):
    """Import a reference module by path (optionally through a source patch: [(old, new), ...])."""
    path = os.path.join(REF, rel)
    if patch is None:
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m
    src = open(path, encoding="utf-8").read()
    for old, new in patch:
        assert src.count(old) == 1, f"patch target not unique in {rel}: {old!r}"
        src = src.replace(old, new)
    m = types.ModuleType(name)
    m.__file__ = path
    sys.modules[name] = m
    exec(compile(src, path, "exec"), m.__dict__)
    return m


# sttn_auto_inpaint.py:245 -- f'...{frame_info['len']}' nests the quote character, legal only from Python 3.12 on
STTN_AUTO_PATCH = [("Total: {frame_info['len']}')", "Total: {frame_info[\"len\"]}')")]
