"""CPU restatement of the reference's ProPainter plugin (test infrastructure only) -- SURVEY.md 8(a) row a13.

Follows backend/inpaint/propainter_inpaint.py: read_mask :32-77 (numpy mask branch), get_ref_index :122-136,
PropainterInpaint.inpaint :190-361, PropainterInpaint.__call__ :363-418, on top of oracle/{raft,rfc,propainter}.py.
fp32 throughout: the reference's CPU path (use_half is forced off on CPU, :146-147).  cv2.cvtColor channel swaps are
restated as slicing; scipy.ndimage.binary_dilation is the reference's own dependency and is called directly.
"""
import numpy as np
import scipy.ndimage
import torch

from vsr_amd.backend.tools.inpaint_tools import get_inpaint_area_by_mask


def dilate_masks(mask, dilation=4):
    """read_mask(mask ndarray, ..., flow_mask_dilates=4, mask_dilates=4): both outputs are the same 4-iteration dilation (:55-70)"""
    m = np.asarray(mask)
    if m.ndim == 3:
        m = m[:, :, 0]
    d = scipy.ndimage.binary_dilation(m, iterations=dilation).astype(np.uint8)
    return d, d.copy()


def get_ref_index(mid_neighbor_id, neighbor_ids, length, ref_stride=10, ref_num=-1):
    ref = []
    if ref_num == -1:
        return [i for i in range(0, length, ref_stride) if i not in neighbor_ids]
    start = max(0, mid_neighbor_id - ref_stride * (ref_num // 2))
    end = min(length, mid_neighbor_id + ref_stride * (ref_num // 2))
    for i in range(start, end, ref_stride):
        if i not in neighbor_ids:
            if len(ref) > ref_num:
                break
            ref.append(i)
    return ref


class PropainterOracle:
    def __init__(self, raft, rfc, gen, sub_video_length=80, raft_iter=20):
        self.raft, self.rfc, self.gen = raft, rfc, gen
        self.sub_video_length = sub_video_length
        self.neighbor_length, self.mask_dilation, self.ref_stride, self.raft_iter = 10, 4, 10, raft_iter

    def inpaint(self, frames_bgr, mask):
        n = len(frames_bgr)
        frames_inp = [np.ascontiguousarray(f[:, :, ::-1]) for f in frames_bgr]          # cv2.COLOR_BGR2RGB
        h, w = frames_inp[0].shape[:2]
        fm, md = dilate_masks(mask, self.mask_dilation)
        frames = torch.from_numpy(np.stack(frames_inp)).permute(0, 3, 1, 2).float().div(255) * 2 - 1
        flow_masks = torch.from_numpy(fm).float()[None, None].repeat(n, 1, 1, 1)
        masks_dilated = torch.from_numpy(md).float()[None, None].repeat(n, 1, 1, 1)
        with torch.no_grad():
            gt_f, gt_b = self.raft.flows_bi(frames, self.raft_iter)                       # chunking (:219-247) does not change pair results
            flow_length = n - 1
            svl = self.sub_video_length
            if flow_length > svl:
                pf, pb = [], []
                for f in range(0, flow_length, svl):
                    s_f, e_f = max(0, f - 5), min(flow_length, f + svl + 5)
                    ps, pe = max(0, f) - s_f, e_f - min(flow_length, f + svl)
                    cf, cb, _, _ = self.rfc.complete_bi(gt_f[s_f:e_f], gt_b[s_f:e_f], flow_masks[s_f:e_f + 1])
                    pf.append(cf[ps:e_f - s_f - pe])
                    pb.append(cb[ps:e_f - s_f - pe])
                pred_f, pred_b = torch.cat(pf), torch.cat(pb)
            else:
                pred_f, pred_b, _, _ = self.rfc.complete_bi(gt_f, gt_b, flow_masks)
            masked_frames = frames * (1 - masks_dilated)
            sip = min(100, svl)
            if n > sip:
                uf, um = [], []
                for f in range(0, n, sip):
                    s_f, e_f = max(0, f - 10), min(n, f + sip + 10)
                    ps, pe = max(0, f) - s_f, e_f - min(n, f + sip)
                    prop, upd = self.gen.img_propagation(masked_frames[s_f:e_f], pred_f[s_f:e_f - 1], pred_b[s_f:e_f - 1],
                                                         masks_dilated[s_f:e_f].clone())
                    sub = frames[s_f:e_f] * (1 - masks_dilated[s_f:e_f]) + prop * masks_dilated[s_f:e_f]
                    uf.append(sub[ps:e_f - s_f - pe])
                    um.append(upd[ps:e_f - s_f - pe])
                updated_frames, updated_masks = torch.cat(uf), torch.cat(um)
            else:
                prop, upd = self.gen.img_propagation(masked_frames, pred_f, pred_b, masks_dilated.clone())
                updated_frames = frames * (1 - masks_dilated) + prop * masks_dilated
                updated_masks = upd
            comp = [None] * n
            stride = self.neighbor_length // 2
            ref_num = svl // self.ref_stride if n > svl else -1
            binary = md[:, :, None].astype(np.uint8)
            for f in range(0, n, stride):
                nb = list(range(max(0, f - stride), min(n, f + stride + 1)))
                ref = get_ref_index(f, nb, n, self.ref_stride, ref_num)
                ids = nb + ref
                l_t = len(nb)
                pred = self.gen.forward(updated_frames[ids], pred_f[nb[:-1]], pred_b[nb[:-1]], masks_dilated[ids], updated_masks[ids], l_t)
                pred = ((pred + 1) / 2).permute(0, 2, 3, 1).numpy() * 255
                for i, idx in enumerate(nb):
                    img = np.array(pred[i]).astype(np.uint8) * binary + frames_inp[idx] * (1 - binary)
                    if comp[idx] is None:
                        comp[idx] = img
                    else:
                        comp[idx] = comp[idx].astype(np.float32) * 0.5 + img.astype(np.float32) * 0.5
                    comp[idx] = comp[idx].astype(np.uint8)
        return [np.ascontiguousarray(c[:, :, ::-1]) for c in comp]                          # cv2.COLOR_RGB2BGR

    def __call__(self, input_frames, input_mask):
        mask = input_mask[:, :, None]
        H, W = mask.shape[:2]
        areas = get_inpaint_area_by_mask(W, H, int(W * 3 / 16), mask, multiple=8)
        out = [f.copy() for f in input_frames]
        comps = [self.inpaint([f[y0:y1, x0:x1, :] for f in input_frames], mask[y0:y1, x0:x1, :]) for (y0, y1, x0, x1) in areas]
        for j, frame in enumerate(out):
            for k, (y0, y1, x0, x1) in enumerate(areas):
                frame[y0:y1, x0:x1, :] = comps[k][j]
        return out
