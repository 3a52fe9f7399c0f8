"""CPU restatement of the reference's ProPainter generator (test infrastructure only) -- SURVEY.md 8(a) row a16.

Follows, in torch-CPU fp32 functional form (batch 1):
  InpaintGenerator.forward / img_propagation        backend/inpaint/video/model/propainter.py:316-378
  Encoder.forward (grouped convs with re-injected x0) :212-224 ; decoder / deconv :227-247,270-277
  BidirectionalPropagation.forward (learnable and not) :104-193, DeformableAlignment.forward :59-72, fbConsistencyCheck :24-33
  flow_warp                                          model/modules/flow_loss_utils.py:6-45
  SoftSplit / SoftComp / FusionFeedForward / SparseWindowAttention / TemporalSparseTransformer(Block)
                                                     model/modules/sparse_transformer.py:7-344
torchvision.ops.deform_conv2d is restated in oracle/deform_conv.py (absent dependency).  Pinned by oracle/make_golden.py
against the reference module run with that same operator restatement (tests/golden/propainter.npz).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .deform_conv import deform_conv2d

KS, ST, PD = (7, 7), (3, 3), (3, 3)          # soft split / composition (propainter.py:280-287)
WIN, POOL, HEADS, DEPTH, HID = (5, 9), (4, 4), 4, 8, 512


def flow_warp(x, flow, mode="bilinear"):
    """x [n,c,h,w], flow [n,h,w,2] (x, y displacement in pixels): grid_sample(align_corners=True, zeros)"""
    n, _, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    grid = torch.stack((gx, gy), 2).to(x) + flow
    gxn = 2.0 * grid[..., 0] / max(w - 1, 1) - 1.0
    gyn = 2.0 * grid[..., 1] / max(h - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack((gxn, gyn), dim=3), mode=mode, padding_mode="zeros", align_corners=True)


def fb_check(flow_fw, flow_bw, alpha1=0.01, alpha2=0.5):
    bw_warped = flow_warp(flow_bw, flow_fw.permute(0, 2, 3, 1))
    diff = flow_fw + bw_warped
    sq = lambda v: torch.sum(torch.square(v), dim=1, keepdim=True)
    return (sq(diff) < alpha1 * (sq(flow_fw) + sq(bw_warped)) + alpha2).to(flow_fw)


def binary_mask(m, th=0.1):
    return (m > th).to(m)


class ProPainterOracle:
    def __init__(self, state_dict):
        self.sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in state_dict.items()}

    def conv(self, x, name, stride=1, padding=1, groups=1):
        return F.conv2d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], stride=stride, padding=padding, groups=groups)

    def lin(self, x, name):
        return F.linear(x, self.sd[name + ".weight"], self.sd[name + ".bias"])

    # ---- propagation (propainter.py:104-193) ------------------------------------------------------
    def deform_align(self, mod, x, cond, flow):
        p = f"feat_prop_module.deform_align.{mod}"
        o = cond
        for i in (0, 2, 4):
            o = F.leaky_relu(self.conv(o, f"{p}.conv_offset.{i}"), 0.1)
        o = self.conv(o, f"{p}.conv_offset.6")
        o1, o2, m = torch.chunk(o, 3, dim=1)
        offset = 3.0 * torch.tanh(torch.cat((o1, o2), dim=1))          # max_residue_magnitude = 3
        offset = offset + flow.flip(1).repeat(1, offset.size(1) // 2, 1, 1)
        return deform_conv2d(x, offset, self.sd[p + ".weight"], self.sd[p + ".bias"], 1, 1, 1, torch.sigmoid(m))

    def propagate(self, x, flows_f, flows_b, mask, learnable, interp):
        """x [t,c,h,w], flows [t-1,2,h,w], mask [t,cm,h,w] -> (backward feats, forward feats, output, forward masks)"""
        t = x.shape[0]
        feats = {"input": [x[i:i + 1] for i in range(t)]}
        masks = {"input": [mask[i:i + 1] for i in range(t)]}
        cache = ["input", "backward_1", "forward_1"]
        for p_i, mod in enumerate(("backward_1", "forward_1")):
            feats[mod], masks[mod] = [], []
            if mod == "backward_1":
                frame_idx = list(range(t))[::-1]
                flow_idx = frame_idx
                f_prop, f_check = flows_f, flows_b
            else:
                frame_idx = list(range(t))
                flow_idx = list(range(-1, t - 1))
                f_prop, f_check = flows_b, flows_f
            prop, mprop = None, None
            for i, idx in enumerate(frame_idx):
                cur, mcur = feats[cache[p_i]][idx], masks[cache[p_i]][idx]
                if i == 0:
                    prop, mprop = cur, mcur
                else:
                    fp = f_prop[flow_idx[i]:flow_idx[i] + 1] if flow_idx[i] >= 0 else f_prop[-1:]
                    fc = f_check[flow_idx[i]:flow_idx[i] + 1] if flow_idx[i] >= 0 else f_check[-1:]
                    valid = fb_check(fp, fc)
                    warped = flow_warp(prop, fp.permute(0, 2, 3, 1), interp)
                    if learnable:
                        cond = torch.cat([cur, warped, fp, valid, mcur], dim=1)
                        prop = self.deform_align(mod, prop, cond, fp)
                        mprop = mcur
                    else:
                        mvalid = binary_mask(flow_warp(mprop, fp.permute(0, 2, 3, 1)))
                        union = binary_mask(mcur * valid * (1 - mvalid))
                        prop = union * warped + (1 - union) * cur
                        mprop = binary_mask(mcur * (1 - (valid * (1 - mvalid))))
                if learnable:
                    b = f"feat_prop_module.backbone.{mod}"
                    y = self.conv(F.leaky_relu(self.conv(torch.cat([cur, prop, mcur], dim=1), b + ".0"), 0.2), b + ".2")
                    prop = prop + y
                feats[mod].append(prop)
                masks[mod].append(mprop)
            if mod == "backward_1":
                feats[mod], masks[mod] = feats[mod][::-1], masks[mod][::-1]
        out_b, out_f = torch.cat(feats["backward_1"], 0), torch.cat(feats["forward_1"], 0)
        if learnable:
            y = self.conv(F.leaky_relu(self.conv(torch.cat([out_b, out_f, mask], dim=1), "feat_prop_module.fuse.0"), 0.2),
                          "feat_prop_module.fuse.2")
            return out_b, out_f, y + x, None
        return out_b, out_f, out_f, torch.cat(masks["forward_1"], 0)

    def img_propagation(self, masked_frames, flows_f, flows_b, masks, interp="nearest"):
        """InpaintGenerator.img_propagation (:316-319): frames [t,3,H,W], flows [t-1,2,H,W], masks [t,1,H,W] -> (frames, updated masks)"""
        with torch.no_grad():
            _, _, out, m = self.propagate(masked_frames, flows_f, flows_b, masks, False, interp)
            return out, m

    # ---- encoder / decoder ---------------------------------------------------------------------------
    def encoder(self, x):
        bt = x.shape[0]
        strides = {0: 2, 4: 2}
        groups = {10: 2, 12: 4, 14: 8}
        out, x0 = x, None
        for i in range(0, 18, 2):
            if i == 8:
                x0 = out
            if i > 8:
                g = [1, 2, 4, 8, 1][(i - 8) // 2]
                h, w = x0.shape[-2:]
                out = torch.cat([x0.view(bt, g, -1, h, w), out.view(bt, g, -1, h, w)], 2).view(bt, -1, h, w)
            out = F.leaky_relu(self.conv(out, f"encoder.layers.{i}", strides.get(i, 1), 1, groups.get(i, 1)), 0.2)
        return out

    def deconv(self, x, name):
        return self.conv(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True), name + ".conv")

    def decoder(self, x):
        y = F.leaky_relu(self.deconv(x, "decoder.0"), 0.2)
        y = F.leaky_relu(self.conv(y, "decoder.2"), 0.2)
        y = F.leaky_relu(self.deconv(y, "decoder.4"), 0.2)
        return self.conv(y, "decoder.6")

    # ---- transformer (sparse_transformer.py) ---------------------------------------------------------
    @staticmethod
    def _fsize(size):
        return tuple(int((size[i] + 2 * PD[i] - (KS[i] - 1) - 1) / ST[i] + 1) for i in range(2))

    def soft_split(self, x, size):
        fh, fw = self._fsize(size)
        feat = F.unfold(x, KS, stride=ST, padding=PD).permute(0, 2, 1)
        return self.lin(feat, "ss.embedding").view(1, -1, fh, fw, HID)

    def soft_comp(self, x, t, size):
        feat = self.lin(x.view(1, -1, HID), "sc.embedding")
        feat = feat.view(t, -1, feat.shape[-1]).permute(0, 2, 1)
        return self.conv(F.fold(feat, size, KS, stride=ST, padding=PD), "sc.bias_conv")

    @staticmethod
    def _windows(x):
        b, t, h, w, c = x.shape
        x = x.view(b, t, h // WIN[0], WIN[0], w // WIN[1], WIN[1], HEADS, c // HEADS)
        return x.permute(0, 2, 4, 6, 1, 3, 5, 7).contiguous()

    def attention(self, blk, x, mask, t_ind):
        p = f"transformers.transformer.{blk}.attention."
        b, t, h, w, c = x.shape
        wh, ww = WIN
        ch = c // HEADS
        nwh, nww = math.ceil(h / wh), math.ceil(w / ww)
        nh, nw = nwh * wh, nww * ww
        if nh > h or nw > w:
            x = F.pad(x, (0, 0, 0, nw - w, 0, nh - h, 0, 0))
            mask = F.pad(mask, (0, 0, 0, nw - w, 0, nh - h, 0, 0))
        q, k, v = self.lin(x, p + "query"), self.lin(x, p + "key"), self.lin(x, p + "value")
        shape = (b, nwh * nww, HEADS, t, wh * ww, ch)
        win_q, win_k, win_v = (self._windows(a).view(shape) for a in (q, k, v))
        eh, ew = (wh + 1) // 2, (ww + 1) // 2
        rolled_k, rolled_v = [], []
        for sh, sw in ((-eh, -ew), (-eh, ew), (eh, -ew), (eh, ew)):
            rolled_k.append(self._windows(torch.roll(k, shifts=(sh, sw), dims=(2, 3))).view(shape))
            rolled_v.append(self._windows(torch.roll(v, shifts=(sh, sw), dims=(2, 3))).view(shape))
        valid = self.sd[p + "valid_ind_rolled"].long()
        win_k = torch.cat((win_k, torch.cat(rolled_k, 4)[:, :, :, :, valid]), dim=4)
        win_v = torch.cat((win_v, torch.cat(rolled_v, 4)[:, :, :, :, valid]), dim=4)
        pool_x = F.conv2d(x.view(b * t, nh, nw, c).permute(0, 3, 1, 2), self.sd[p + "pool_layer.weight"], self.sd[p + "pool_layer.bias"],
                          stride=POOL, groups=c)
        ph, pw = pool_x.shape[-2:]
        pool_x = pool_x.permute(0, 2, 3, 1).view(b, t, ph, pw, c)
        for name, lst in (("key", "k"), ("value", "v")):
            pk = self.lin(pool_x, p + name).unsqueeze(1).repeat(1, nwh * nww, 1, 1, 1, 1)
            pk = pk.view(b, nwh * nww, t, ph, pw, HEADS, ch).permute(0, 1, 5, 2, 3, 4, 6).contiguous().view(b, nwh * nww, HEADS, t, ph * pw, ch)
            if lst == "k":
                win_k = torch.cat((win_k, pk), dim=4)
            else:
                win_v = torch.cat((win_v, pk), dim=4)
        out = torch.zeros_like(win_q)
        lt = mask.size(1)
        wm = F.max_pool2d(mask.view(b * lt, nh, nw), WIN, WIN).view(b, lt, nwh * nww).sum(dim=1)
        scale = 1.0 / math.sqrt(ch)
        mi = wm[0].nonzero(as_tuple=False).view(-1)
        if len(mi) > 0:
            qt = win_q[0, mi].view(len(mi), HEADS, t * wh * ww, ch)
            kt = win_k[0, mi][:, :, t_ind.view(-1)].reshape(len(mi), HEADS, -1, ch)
            vt = win_v[0, mi][:, :, t_ind.view(-1)].reshape(len(mi), HEADS, -1, ch)
            att = F.softmax((qt @ kt.transpose(-2, -1)) * scale, dim=-1)
            out[0, mi] = (att @ vt).view(-1, HEADS, t, wh * ww, ch)
        ui = (wm[0] == 0).nonzero(as_tuple=False).view(-1)
        qs, ks_, vs = win_q[0, ui], win_k[0, ui, :, :, :wh * ww], win_v[0, ui, :, :, :wh * ww]
        att = F.softmax((qs @ ks_.transpose(-2, -1)) * scale, dim=-1)
        out[0, ui] = att @ vs
        out = out.view(b, nwh, nww, HEADS, t, wh, ww, ch).permute(0, 4, 1, 5, 2, 6, 3, 7).contiguous().view(b, t, nh, nw, c)
        return self.lin(out[:, :, :h, :w, :], p + "proj")

    def ffn(self, blk, x, size):
        p = f"transformers.transformer.{blk}.mlp."
        fh, fw = self._fsize(size)
        nv = fh * fw
        x = self.lin(x, p + "fc1.0")
        b, n, c = x.shape
        norm = F.fold(x.new_ones(b, n, 49).view(-1, nv, 49).permute(0, 2, 1), size, KS, padding=PD, stride=ST)
        y = F.fold(x.view(-1, nv, c).permute(0, 2, 1), size, KS, padding=PD, stride=ST)
        y = F.unfold(y / norm, KS, padding=PD, stride=ST).permute(0, 2, 1).contiguous().view(b, n, c)
        return self.lin(F.gelu(y), p + "fc2.1")

    def transformer(self, x, size, lmask, t_dilation=2):
        T = x.size(1)
        t_inds = [torch.arange(i, T, t_dilation) for i in range(t_dilation)] * (DEPTH // t_dilation)
        for i in range(DEPTH):
            p = f"transformers.transformer.{i}."
            B, _, H, W, C = x.shape
            y = F.layer_norm(x, (C,), self.sd[p + "norm1.weight"], self.sd[p + "norm1.bias"])
            x = x + self.attention(i, y, lmask, t_inds[i])
            y = F.layer_norm(x, (C,), self.sd[p + "norm2.weight"], self.sd[p + "norm2.bias"])
            x = x + self.ffn(i, y.view(B, T * H * W, C), size).view(B, T, H, W, C)
        return x

    # ---- InpaintGenerator.forward (:321-378), eval mode ----------------------------------------------
    def forward(self, masked_frames, flows_f, flows_b, masks_in, masks_updated, l_t, interp="bilinear", t_dilation=2):
        """masked_frames [t,3,H,W], flows [l_t-1,2,H,W], masks [t,1,H,W] -> tanh output [l_t,3,H,W]"""
        with torch.no_grad():
            t, _, H, W = masked_frames.shape
            enc = self.encoder(torch.cat([masked_frames, masks_in, masks_updated], dim=1))
            c, h, w = enc.shape[1:]
            local, ref = enc[:l_t], enc[l_t:]
            ds_f = F.interpolate(flows_f, scale_factor=1 / 4, mode="bilinear", align_corners=False) / 4.0
            ds_b = F.interpolate(flows_b, scale_factor=1 / 4, mode="bilinear", align_corners=False) / 4.0
            ds_mask_in = F.interpolate(masks_in, scale_factor=1 / 4, mode="nearest")
            ds_mask_upd = F.interpolate(masks_updated[:l_t], scale_factor=1 / 4, mode="nearest")
            mask_pool = F.max_pool2d(ds_mask_in[:l_t], KS, ST, PD)
            prop_mask = torch.cat([ds_mask_in[:l_t], ds_mask_upd], dim=1)
            _, _, local, _ = self.propagate(local, ds_f, ds_b, prop_mask, True, interp)
            enc = torch.cat((local, ref), dim=0)
            tok = self.soft_split(enc, (h, w))
            tok = self.transformer(tok, (h, w), mask_pool.permute(0, 2, 3, 1)[None].contiguous(), t_dilation)
            enc = enc + self.soft_comp(tok, t, (h, w))
            return torch.tanh(self.decoder(enc[:l_t]))
