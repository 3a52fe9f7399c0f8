"""LaMa plugin with the reference's signature, running on the MI355X engine.

Mirrors backend/inpaint/lama_inpaint.py:
  LamaInpaint(device, model_path)                           :12-15
      inpaint(image, mask) -> uint8 HxWx3                    :17-28   (single image, whole frame: main.py:220,233,364)
      _inpaint_batch(images, masks) -> [uint8 hxWx3]         :30-66   (mini-batches of 4)
      __call__(input_frames, input_mask) -> frames           :68-114  (strips of height int(W*3/16), whole strip overwritten)
The padding to multiples of 8, the {0,1} mask, clip(x*255).astype(uint8) and the crop (lama_util.py:12-80) happen inside the
engine's kernels; this file only moves frames to HBM and back.  `model_path`: a state_dict ({"model.1.ffc...": array}; with or
without the exported module's `generator.` prefix), an .npz / .pth holding one -- or big-lama.pt itself when it can be opened
with torch.jit.load (its parameters are read, the module is never executed).  There is no CPU path.
"""
import numpy as np
import torch

from ..tools.inpaint_tools import get_inpaint_area_by_mask
from ...engine import LamaEngine
from .sttn_auto_inpaint import _device_index


def _load_lama_state_dict(model_path):
    if isinstance(model_path, dict):
        return model_path
    from ..tools.common_tools import checkpoint_path

    path = checkpoint_path(str(model_path))                                  # big-lama.pt ships in 50 MB parts (model_config.py:24)
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    try:
        module = torch.jit.load(path, map_location="cpu")                    # lama_inpaint.py:13
        sd = module.state_dict()
    except Exception:
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
    out = {}
    for k, v in sd.items():
        if ".model." in "." + k or k.startswith("model."):
            k2 = k[k.index("model."):] if not k.startswith("model.") else k
            out[k2] = v
    if not out:
        raise KeyError(f"{path}: no generator entries ('model.N...' or 'generator.model.N...') among its {len(sd)} tensors: {list(sd)[:4]}")
    return out


class LamaInpaint:
    accepts_device_frames = True      # __call__ also takes a uint8 [n,H,W,3] device tensor and inpaints it in place (tools/resident.py)

    mini_batch_size = 4                                                      # lama_inpaint.py:37

    def __init__(self, device="cuda:0", model_path="big-lama.pt"):
        self.device = device
        self.engine = LamaEngine(_load_lama_state_dict(model_path), device=_device_index(device))
        self._model_path = model_path

    def clone(self):
        """a second instance on the same device from the same weights: its own engine and workspace (tools/batch_lanes.py)"""
        return LamaInpaint(self.device, self._model_path)

    def close(self):
        self.engine.close()

    def inpaint(self, image, mask):
        img = np.ascontiguousarray(np.array(image))
        msk = np.array(mask)
        if msk.ndim == 3:
            msk = msk[:, :, 0]
        dev = self.engine.device
        out = self.engine.inpaint(torch.from_numpy(img)[None].to(dev), torch.from_numpy(np.ascontiguousarray(msk)).to(dev))
        return out[0].cpu().numpy()

    def _inpaint_batch(self, images, masks):
        if len(images) == 1:
            return [self.inpaint(images[0], masks[0])]
        dev = self.engine.device
        imgs = torch.from_numpy(np.ascontiguousarray(np.stack(images))).to(dev)
        msks = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(m).reshape(m.shape[0], m.shape[1]) for m in masks]))).to(dev)
        out = torch.empty_like(imgs)
        for s in range(0, len(images), self.mini_batch_size):
            e = min(s + self.mini_batch_size, len(images))
            self.engine.inpaint(imgs[s:e], msks[s:e], out=out[s:e])
        res = out.cpu().numpy()
        return [res[i] for i in range(res.shape[0])]

    def __call__(self, input_frames, input_mask):
        mask = input_mask[:, :, None]
        H_ori, W_ori = mask.shape[:2]
        split_h = int(W_ori * 3 / 16)
        inpaint_area = get_inpaint_area_by_mask(W_ori, H_ori, split_h, mask)
        resident = isinstance(input_frames, torch.Tensor)       # a uint8 [n,H,W,3] device tensor (tools/resident.py): in place
        if not inpaint_area or len(input_frames) == 0:
            return input_frames if resident else [f.copy() for f in input_frames]
        dev = self.engine.device
        frames = input_frames if resident else torch.from_numpy(np.ascontiguousarray(np.stack(input_frames))).to(dev)
        dmask = torch.from_numpy(np.ascontiguousarray(input_mask)).to(dev)
        n = frames.shape[0]
        # The reference crops every strip from the ORIGINAL frames, runs them all, and only then writes them back in order
        # (:88-106).  Strips of two subtitle groups closer than split_h overlap; in place, the second strip would be fed the
        # first one's output in the shared rows.  Overlap is rare: only then is a snapshot of the frames kept.
        spans = sorted((a[0], a[1]) for a in inpaint_area)
        overlap = any(spans[i][1] > spans[i + 1][0] for i in range(len(spans) - 1))
        source = frames.clone() if overlap else frames
        for y0, y1, _, _ in inpaint_area:                                    # full-width strips: row slices are contiguous rows
            strip_mask = dmask[y0:y1].contiguous()
            if n == 1:                                                       # :32-33 -> inpaint(): same arithmetic
                self.engine.inpaint(source[:, y0:y1], strip_mask, out=frames[:, y0:y1])
                continue
            for s in range(0, n, self.mini_batch_size):
                e = min(s + self.mini_batch_size, n)
                self.engine.inpaint(source[s:e, y0:y1], strip_mask, out=frames[s:e, y0:y1])
        if resident:
            return frames
        out = frames.cpu().numpy()
        return [out[i] for i in range(n)]
