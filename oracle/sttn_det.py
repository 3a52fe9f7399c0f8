"""Restatement of the sttn-det wrapper (oracle; see __init__.py).

Follows backend/inpaint/sttn_det_inpaint.py: STTNDetInpaint.__call__ (:38-99) and .inpaint (:124-174).
Differences from sttn-auto that change pixels (SURVEY.md Appendix B 1,7,8):
  * the mask strip is resized with the frames (cv2.resize of the 0/255 uint8 mask -> 0..255 values);
  * encoder input is frames*(1 - (mask/255 > 0.5)) -- i.e. mask >= 128 -- (:134,143);
  * the attention mask handed to the network has no effect (network_sttn.py:149);
  * model-resolution blend pred*b + frame*(1-b) with b = mask > 0.5 ON 0..255 DATA, i.e. any non-zero (:132,168);
    `frames[idx]` is RGB at that point because Stack() replaced the list items in place (sttn_utils.py:70-75);
  * the whole strip is overwritten with the up-scaled composite (:93), strip height int(W*5/18) (:48-51).
"""
import numpy as np
import torch

from . import cv2_restate as cv2r
from .sttn_auto import get_inpaint_area_by_mask
from .sttn_net import MODEL_SIZE, SttnNet


class STTNDetOracle:
    def __init__(self, state_dict, neighbor_stride=5, ref_length=10):
        self.net = SttnNet(state_dict, "det")
        self.model_input_width, self.model_input_height = MODEL_SIZE["det"]
        self.neighbor_stride = neighbor_stride
        self.ref_length = ref_length

    def get_ref_index(self, neighbor_ids, length):
        return [i for i in range(0, length, self.ref_length) if i not in neighbor_ids]

    def inpaint(self, frames, masks):
        """frames: list of 240x432x3 uint8 BGR; masks: list of 240x432 uint8 (resized 0/255 mask)."""
        frame_length = len(frames)
        rgb = [f[:, :, ::-1] for f in frames]                                   # Stack(): BGR -> RGB, in place in the reference
        feats = torch.from_numpy(np.ascontiguousarray(np.stack(rgb))).permute(0, 3, 1, 2).float().div(255) * 2 - 1
        binary_masks = [np.expand_dims((np.array(m) > 0.5).astype(np.uint8), 2) for m in masks]
        masks_tensor = (torch.from_numpy(np.stack(masks)).unsqueeze(1).float().div(255) > 0.5).float()
        comp_frames = [None] * frame_length
        with torch.no_grad():
            feats = self.net.encoder(feats * (1 - masks_tensor))
            for f in range(0, frame_length, self.neighbor_stride):
                neighbor_ids = list(range(max(0, f - self.neighbor_stride), min(frame_length, f + self.neighbor_stride + 1)))
                ref_ids = self.get_ref_index(neighbor_ids, frame_length)
                pred_feat = self.net.infer(feats[neighbor_ids + ref_ids])        # the mask argument is a no-op
                pred_img = torch.tanh(self.net.decoder(pred_feat[:len(neighbor_ids)]))
                pred_img = (pred_img + 1) / 2
                pred_img = pred_img.cpu().permute(0, 2, 3, 1).numpy() * 255
                for i, idx in enumerate(neighbor_ids):
                    img = pred_img[i].astype(np.uint8) * binary_masks[idx] + rgb[idx] * (1 - binary_masks[idx])
                    if comp_frames[idx] is None:
                        comp_frames[idx] = img
                    else:
                        comp_frames[idx] = comp_frames[idx].astype(np.float32) * 0.5 + img.astype(np.float32) * 0.5
        return comp_frames

    def split_height(self, W_ori, H_ori):
        return int(H_ori * 5 / 9) if H_ori > W_ori else int(W_ori * 5 / 18)

    def __call__(self, input_frames, input_mask):
        """input_mask: HxW uint8 in {0,255} (create_mask output, no thresholding in this plugin)."""
        mask = input_mask[:, :, None]
        H_ori, W_ori = mask.shape[:2]
        split_h = self.split_height(W_ori, H_ori)
        inpaint_area = get_inpaint_area_by_mask(W_ori, H_ori, split_h, mask)
        frames_hr = [f.copy() for f in input_frames]
        size = (self.model_input_width, self.model_input_height)
        comps = {}
        for k, area in enumerate(inpaint_area):
            fs = [cv2r.resize_linear(f[area[0]:area[1], :, :], size) for f in frames_hr]
            ms = [cv2r.resize_linear(mask[area[0]:area[1], :, :], size)[:, :, 0] for _ in frames_hr]
            comps[k] = self.inpaint(fs, ms)
        for j, frame in enumerate(frames_hr):
            for k, area in enumerate(inpaint_area):
                comp = cv2r.resize_linear(comps[k][j], (W_ori, split_h))
                frame[area[0]:area[1], :, :] = comp.astype(np.uint8)[:, :, ::-1]
        return frames_hr
