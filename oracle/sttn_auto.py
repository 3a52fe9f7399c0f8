"""Restatement of the sttn-auto wrapper logic around the network (oracle; see __init__.py).

Follows backend/inpaint/sttn_auto_inpaint.py: STTNInpaint.inpaint (:122-164),
get_ref_index (:107-120), STTNInpaint.__call__ (:43-97) and the per-chunk body of
STTNAutoInpaint.__call__ (:242-317); pre-processing from backend/inpaint/utils/sttn_utils.py
(Stack :66-86, ToTorchFormatTensor :89-112).
"""
import numpy as np
import torch

from . import cv2_restate as cv2r
from .sttn_net import MODEL_SIZE, SttnNet


class STTNInpaintOracle:
    def __init__(self, state_dict, variant="auto", neighbor_stride=5, ref_length=10):
        self.net = SttnNet(state_dict, variant)
        self.model_input_width, self.model_input_height = MODEL_SIZE[variant]
        self.neighbor_stride = neighbor_stride     # config.sttnNeighborStride (backend/config.py)
        self.ref_length = ref_length               # config.sttnReferenceLength

    # sttn_auto_inpaint.py:107-120
    def get_ref_index(self, neighbor_ids, length):
        return [i for i in range(0, length, self.ref_length) if i not in neighbor_ids]

    def window_schedule(self, frame_length):
        out = []
        for f in range(0, frame_length, self.neighbor_stride):
            neighbor_ids = list(range(max(0, f - self.neighbor_stride),
                                      min(frame_length, f + self.neighbor_stride + 1)))
            out.append((neighbor_ids, self.get_ref_index(neighbor_ids, frame_length)))
        return out

    @staticmethod
    def to_tensors(frames):
        """Stack (BGR->RGB, stack) + ToTorchFormatTensor (/255) -> [T,3,H,W] float32 in [0,1]."""
        arr = np.stack([f[:, :, ::-1] for f in frames], axis=0)            # T,H,W,C (RGB)
        return torch.from_numpy(np.ascontiguousarray(arr)).permute(0, 3, 1, 2).contiguous().float().div(255)

    # sttn_auto_inpaint.py:122-164
    def inpaint(self, frames):
        frame_length = len(frames)
        feats = self.to_tensors(frames).unsqueeze(0) * 2 - 1
        comp_frames = [None] * frame_length
        with torch.no_grad():
            feats = self.net.encoder(feats.view(frame_length, 3, self.model_input_height, self.model_input_width))
            for neighbor_ids, ref_ids in self.window_schedule(frame_length):
                pred_feat = self.net.infer(feats[neighbor_ids + ref_ids, :, :, :])
                pred_img = torch.tanh(self.net.decoder(pred_feat[:len(neighbor_ids), :, :, :]))
                pred_img = (pred_img + 1) / 2
                pred_img = pred_img.cpu().permute(0, 2, 3, 1).numpy() * 255
                for i, idx in enumerate(neighbor_ids):
                    img = pred_img[i].astype(np.uint8)
                    if comp_frames[idx] is None:
                        comp_frames[idx] = img
                    else:
                        comp_frames[idx] = comp_frames[idx].astype(np.float32) * 0.5 + img.astype(np.float32) * 0.5
        return comp_frames

    def blend_strip(self, frame, comp, mask01, area, W_ori, split_h):
        """sttn_auto_inpaint.py:312-315 for one frame and one area (frame is modified in place)."""
        ymin, ymax = area[0], area[1]
        comp = cv2r.resize_linear(comp, (W_ori, split_h))
        comp = comp.astype(np.uint8)[:, :, ::-1]                           # cvtColor(BGR2RGB) = channel swap
        mask_area = mask01[ymin:ymax, :]
        frame[ymin:ymax, :, :] = mask_area * comp + (1 - mask_area) * frame[ymin:ymax, :, :]

    def __call__(self, input_frames, input_mask):
        """STTNInpaint.__call__ (:43-97), the generic list-in / list-out plugin contract: threshold the 0/255 mask, find the
        strips, then exactly the chunk body below over all frames."""
        mask01 = cv2r.threshold_binary(input_mask, 127, 1)[:, :, None]
        H_ori, W_ori = mask01.shape[:2]
        inpaint_area = get_inpaint_area_by_mask(W_ori, H_ori, int(W_ori * 3 / 16), mask01)
        if not inpaint_area:
            return [f.copy() for f in input_frames]
        return self.chunk(input_frames, mask01, inpaint_area)

    def chunk(self, frames_hr, mask01, inpaint_area, sel=None):
        """Body of the chunk loop of STTNAutoInpaint.__call__ (:242-317): returns the written frames.

        mask01: HxWx1 uint8 in {0,1} (after cv2.threshold(.,127,1)); sel: indices of frames inside the
        A/B sections (None = all)."""
        H_ori, W_ori = mask01.shape[:2]
        split_h = int(W_ori * 3 / 16)
        frames_hr = [f.copy() for f in frames_hr]
        sel = list(range(len(frames_hr))) if sel is None else list(sel)
        frames = {k: [] for k in range(len(inpaint_area))}
        for j in sel:
            for k, area in enumerate(inpaint_area):
                crop = frames_hr[j][area[0]:area[1], :, :]
                frames[k].append(cv2r.resize_linear(crop, (self.model_input_width, self.model_input_height)))
        comps = {k: (self.inpaint(frames[k]) if frames[k] else []) for k in range(len(inpaint_area))}
        for ci, j in enumerate(sel):
            for k, area in enumerate(inpaint_area):
                self.blend_strip(frames_hr[j], comps[k][ci], mask01, area, W_ori, split_h)
        return frames_hr


# --------------------------------------------------------------------------------------------
# backend/tools/inpaint_tools.py restated (host bookkeeping that decides which pixels are inpainted)
# --------------------------------------------------------------------------------------------
SUBTITLE_AREA_DEVIATION_PIXEL = 10   # backend/config.py:61


def create_mask(size, coords_list):
    """tools/inpaint_tools.py:31-47: filled boxes grown by 10 px, low side clamped at 0 only."""
    mask = np.zeros(size, dtype=np.uint8)
    for xmin, xmax, ymin, ymax in (coords_list or []):
        x1 = max(xmin - SUBTITLE_AREA_DEVIATION_PIXEL, 0)
        y1 = max(ymin - SUBTITLE_AREA_DEVIATION_PIXEL, 0)
        cv2r.rectangle_filled(mask, (x1, y1), (xmax + SUBTITLE_AREA_DEVIATION_PIXEL, ymax + SUBTITLE_AREA_DEVIATION_PIXEL), 255)
    return mask


def batch_generator(data, max_batch_size):
    """tools/inpaint_tools.py:7-29 (shrinks the batch while n % bs < bs/2, remainder 0 included)."""
    n = len(data)
    bs = max_batch_size
    nb = n // bs
    while n % bs < bs / 2.0 and bs > 1:
        bs -= 1
        nb = n // bs
    for i in range(nb):
        yield data[i * bs:(i + 1) * bs]
    if nb * bs < n:
        yield data[nb * bs:]


def get_inpaint_area_by_mask(W, H, h, mask, multiple=1):
    """tools/inpaint_tools.py:49-242: 8-connected islands (area >= 10) sorted by centroid y, merged
    while connected and the span stays <= h, each group turned into a full-width strip of height h."""
    areas = []
    if np.all(mask == 0):
        return areas
    binary = (mask > 0).astype(np.uint8) * 255
    if binary.ndim == 3:
        binary = binary[:, :, 0]
    n, _, stats, cents = cv2r.connected_components_with_stats(binary, 8)
    islands = []
    for i in range(1, n):
        left, top, width, height, area = (int(v) for v in stats[i])
        if area < 10:
            continue
        islands.append((top, top + height, int(cents[i][1]), area, i))
    if not islands:
        return areas
    islands.sort(key=lambda t: t[2])
    groups, cur = [], [islands[0]]
    for isl in islands[1:]:
        lo = min(t[0] for t in cur)
        hi = max(t[1] for t in cur)
        new_lo, new_hi = min(lo, isl[0]), max(hi, isl[1])
        connected = bool(np.any(binary[hi:isl[0], :] > 0)) if hi < isl[0] else True
        if new_hi - new_lo <= h and connected:
            cur.append(isl)
        else:
            groups.append(cur)
            cur = [isl]
    groups.append(cur)

    def clamp_bottom(ymin):
        ymax = ymin + h
        if ymax > H:
            ymax = H
            ymin = max(0, H - h)
        return ymin, ymax

    for grp in groups:
        lo = min(t[0] for t in grp)
        hi = max(t[1] for t in grp)
        center = sum(t[2] for t in grp) // len(grp)
        half = h // 2
        ymin, ymax = clamp_bottom(max(0, center - half))
        if ymin > lo or ymax < hi:
            if hi - lo <= h:
                ymin, ymax = clamp_bottom(lo)
            else:
                ymin, ymax = clamp_bottom(max(0, (lo + hi) // 2 - half))
        xmin, xmax = 0, W
        if multiple > 1:
            height = ymax - ymin
            rem = height % multiple
            if rem != 0:
                adj = multiple - rem
                cy = (ymin + ymax) / 2
                if ymin - adj / 2 >= 0 and ymax + adj / 2 <= H:
                    ymin = int(cy - height / 2 - adj / 2)
                    ymax = int(cy + height / 2 + adj / 2)
                elif height > multiple:
                    ymin = int(cy - (height - rem) / 2)
                    ymax = int(cy + (height - rem) / 2)
                elif ymax + adj <= H:
                    ymax += adj
                elif ymin - adj >= 0:
                    ymin -= adj
            width = xmax - xmin
            rem_w = width % multiple
            if rem_w != 0:
                cx = (xmin + xmax) / 2
                xmin = int(cx - (width - rem_w) / 2)
                xmax = int(cx + (width - rem_w) / 2)
        area = (int(ymin), int(ymax), int(xmin), int(xmax))
        if area not in areas:
            areas.append(area)
    return areas


def calculate_psnr(img1, img2):
    """backend/inpaint/video/core/metrics.py:20-36: 20*log10(255/sqrt(MSE)), inf when identical."""
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse == 0:
        return float("inf")
    return float(20 * np.log10(255.0 / np.sqrt(mse)))
