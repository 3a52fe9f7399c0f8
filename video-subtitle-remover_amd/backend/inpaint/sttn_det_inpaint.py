"""sttn-det plugin with the reference's signature, running on the MI355X engine.

Mirrors backend/inpaint/sttn_det_inpaint.py:
  STTNDetInpaint(device, model_path)                     :23-36
      __call__(input_frames, input_mask) -> frames        :38-99   (the generic plugin contract of main.py:326)
      inpaint(frames, masks) -> comp frames               :124-174
      get_ref_index(neighbor_ids, length)                 :108-122
The caller (SubtitleRemover.video_inpaint, main.py:323-331) passes batches of <= 50 frames from batch_generator
and the 0/255 mask of the interval; this class moves the batch to HBM, makes one C call and returns fresh arrays.
"""
import numpy as np
import torch

from ..config import config
from ..tools.inpaint_tools import get_inpaint_area_by_mask
from ...engine import SttnEngine
from .sttn_auto_inpaint import _device_index, _load_state_dict


class STTNDetInpaint:
    accepts_device_frames = True      # __call__ also takes a uint8 [n,H,W,3] device tensor and inpaints it in place (tools/resident.py)

    def __init__(self, device, model_path):
        self.device = device
        self.neighbor_stride = config.sttnNeighborStride.value
        self.ref_length = config.sttnReferenceLength.value
        self.engine = SttnEngine(_load_state_dict(model_path), "det", device=_device_index(device),
                                 neighbor_stride=self.neighbor_stride, ref_length=self.ref_length)
        self.model_input_width, self.model_input_height = 432, 240
        self._model_path = model_path

    def clone(self):
        """a second instance on the same device from the same checkpoint: its own engine and workspace (tools/batch_lanes.py)"""
        return STTNDetInpaint(self.device, self._model_path)

    def __call__(self, input_frames, input_mask):
        """input_frames: the reference's list of HxWx3 uint8 BGR arrays (fresh arrays come back), or -- the HBM-resident loop of
        main.SubtitleRemover, tools/resident.py -- a contiguous uint8 [n,H,W,3] device tensor, which is inpainted IN PLACE and
        returned."""
        mask = input_mask[:, :, None]
        H_ori, W_ori = mask.shape[:2]
        split_h = int(H_ori * 5 / 9) if H_ori > W_ori else int(W_ori * 5 / 18)
        inpaint_area = get_inpaint_area_by_mask(W_ori, H_ori, split_h, mask)
        if isinstance(input_frames, torch.Tensor):
            if inpaint_area and input_frames.shape[0]:
                dmask = torch.from_numpy(np.ascontiguousarray(input_mask)).to(input_frames.device, non_blocking=True)
                self.engine.det_batch(input_frames, dmask, inpaint_area, mask_host=input_mask)
            return input_frames
        if not inpaint_area or len(input_frames) == 0:
            return [f.copy() for f in input_frames]
        dev = self.engine.device
        frames = torch.from_numpy(np.ascontiguousarray(np.stack(input_frames))).to(dev, non_blocking=True)
        dmask = torch.from_numpy(np.ascontiguousarray(input_mask)).to(dev, non_blocking=True)
        self.engine.det_batch(frames, dmask, inpaint_area, mask_host=input_mask)
        out = frames.cpu().numpy()
        return [out[i] for i in range(out.shape[0])]

    @staticmethod
    def read_mask(path):
        from PIL import Image

        img = np.array(Image.open(path).convert("L"))
        return (img > 127).astype(np.uint8)[:, :, None]

    def get_ref_index(self, neighbor_ids, length):
        return [i for i in range(0, length, self.ref_length) if i not in neighbor_ids]

    def inpaint(self, frames, masks):
        dev = self.engine.device
        d = torch.from_numpy(np.ascontiguousarray(np.stack(frames))).to(dev)
        m = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(x) for x in masks]))).to(dev)
        comp, counts = self.engine.det_inpaint(d, m)
        comp = comp.cpu().numpy()
        return [comp[i].astype(np.uint8) if counts[i] == 1 else comp[i] for i in range(len(frames))]
