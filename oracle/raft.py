"""CPU restatement of the reference's RAFT optical-flow network (test infrastructure only -- never imported by the
product path).  SURVEY.md section 8(a) row a14.

Follows, in torch-CPU fp32 functional form:
  RAFT.forward / upsample_flow / initialize_flow      backend/inpaint/video/raft/raft.py:62-146
  BasicEncoder / ResidualBlock                         raft/extractor.py:6-58,118-192   (fnet: InstanceNorm, cnet: BatchNorm eval)
  CorrBlock (all-pairs volume, 4-level pyramid, lookup) raft/corr.py:12-60  (incl. the (dy,dx)-added-to-(x,y) window order, :37-43)
  BasicMotionEncoder / SepConvGRU / FlowHead / mask     raft/update.py:6-14,33-60,79-98,115-137
  bilinear_sampler / coords_grid                        raft/utils/utils.py:57-76
  RAFT_bi.forward (all consecutive pairs, both directions) backend/inpaint/video/model/modules/flow_comp_raft.py:39-55

Pinned by oracle/make_golden.py against the reference module itself (tests/golden/raft.npz).
"""
import numpy as np
import torch
import torch.nn.functional as F

LEVELS, RADIUS, HDIM, CDIM = 4, 4, 128, 128


class RaftOracle:
    def __init__(self, state_dict):
        self.sd = {}
        for k, v in state_dict.items():
            k = k[7:] if k.startswith("module.") else k          # DataParallel checkpoint (flow_comp_raft.py:17-19)
            self.sd[k] = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))

    # ---- building blocks ------------------------------------------------------------------
    def conv(self, x, name, stride=1, padding=0):
        return F.conv2d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], stride=stride, padding=padding)

    def norm(self, x, name, kind):
        if kind == "instance":                                   # nn.InstanceNorm2d defaults: no affine, no running stats
            return F.instance_norm(x, eps=1e-5)
        g = self.sd
        return F.batch_norm(x, g[name + ".running_mean"], g[name + ".running_var"], g[name + ".weight"], g[name + ".bias"],
                            training=False, eps=1e-5)

    def res_block(self, x, p, kind, stride):
        y = torch.relu(self.norm(self.conv(x, p + "conv1", stride, 1), p + "norm1", kind))
        y = torch.relu(self.norm(self.conv(y, p + "conv2", 1, 1), p + "norm2", kind))
        if stride != 1:
            x = self.norm(self.conv(x, p + "downsample.0", stride, 0), p + "norm3", kind)
        return torch.relu(x + y)

    def encoder(self, x, prefix, kind):
        x = torch.relu(self.norm(self.conv(x, prefix + "conv1", 2, 3), prefix + "norm1", kind))
        for li, stride in ((1, 1), (2, 2), (3, 2)):
            x = self.res_block(x, f"{prefix}layer{li}.0.", kind, stride)
            x = self.res_block(x, f"{prefix}layer{li}.1.", kind, 1)
        return self.conv(x, prefix + "conv2")

    @staticmethod
    def corr_pyramid(f1, f2):
        b, d, h, w = f1.shape
        vol = torch.matmul(f1.reshape(b, d, h * w).transpose(1, 2), f2.reshape(b, d, h * w)) / np.sqrt(np.float32(d))
        vol = vol.reshape(b * h * w, 1, h, w)
        pyr = [vol]
        for _ in range(LEVELS - 1):
            vol = F.avg_pool2d(vol, 2, stride=2)
            pyr.append(vol)
        return pyr

    @staticmethod
    def lookup(pyr, coords):
        """coords [b,2,h,w] (x,y) -> [b, 4*81, h, w]; window offset (i,j) adds offs[i] to x and offs[j] to y, and the
        channel index is 9*i + j (the reference's meshgrid(dy,dx) stacked onto (x,y) coordinates, corr.py:37-43)."""
        b, _, h, w = coords.shape
        r = RADIUS
        offs = torch.linspace(-r, r, 2 * r + 1)
        di, dj = torch.meshgrid(offs, offs, indexing="ij")
        c = coords.permute(0, 2, 3, 1).reshape(b * h * w, 1, 1, 2)
        out = []
        for lvl, vol in enumerate(pyr):
            hh, ww = vol.shape[-2:]
            x = c[..., 0] / 2 ** lvl + di[None]
            y = c[..., 1] / 2 ** lvl + dj[None]
            grid = torch.stack([2 * x / (ww - 1) - 1, 2 * y / (hh - 1) - 1], dim=-1)
            s = F.grid_sample(vol, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
            out.append(s.reshape(b, h, w, -1))
        return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous()

    def update(self, net, inp, corr, flow, want_mask):
        u = "update_block."
        cor = torch.relu(self.conv(corr, u + "encoder.convc1"))
        cor = torch.relu(self.conv(cor, u + "encoder.convc2", 1, 1))
        flo = torch.relu(self.conv(flow, u + "encoder.convf1", 1, 3))
        flo = torch.relu(self.conv(flo, u + "encoder.convf2", 1, 1))
        mot = torch.relu(self.conv(torch.cat([cor, flo], 1), u + "encoder.conv", 1, 1))
        x = torch.cat([inp, mot, flow], 1)
        for tag, pad in (("1", (0, 2)), ("2", (2, 0))):
            hx = torch.cat([net, x], 1)
            z = torch.sigmoid(self.conv(hx, u + "gru.convz" + tag, 1, pad))
            r = torch.sigmoid(self.conv(hx, u + "gru.convr" + tag, 1, pad))
            q = torch.tanh(self.conv(torch.cat([r * net, x], 1), u + "gru.convq" + tag, 1, pad))
            net = (1 - z) * net + z * q
        delta = self.conv(torch.relu(self.conv(net, u + "flow_head.conv1", 1, 1)), u + "flow_head.conv2", 1, 1)
        mask = None
        if want_mask:
            mask = 0.25 * self.conv(torch.relu(self.conv(net, u + "mask.0", 1, 1)), u + "mask.2")
        return net, mask, delta

    @staticmethod
    def upsample(flow, mask):
        n, _, h, w = flow.shape
        m = torch.softmax(mask.reshape(n, 1, 9, 8, 8, h, w), dim=2)
        nb = F.unfold(8 * flow, [3, 3], padding=1).reshape(n, 2, 9, 1, 1, h, w)
        up = torch.sum(m * nb, dim=2)                               # [n,2,8,8,h,w]
        return up.permute(0, 1, 4, 2, 5, 3).reshape(n, 2, 8 * h, 8 * w)

    # ---- the network -----------------------------------------------------------------------
    def forward(self, image1, image2, iters=20):
        """image1/2: [n,3,H,W] fp32 in [-1,1], H and W multiples of 8 -> (flow at 1/8 [n,2,H/8,W/8], flow [n,2,H,W])."""
        with torch.no_grad():
            n = image1.shape[0]
            f = self.encoder(torch.cat([image1, image2], 0), "fnet.", "instance")
            pyr = self.corr_pyramid(f[:n].float(), f[n:].float())
            c = self.encoder(image1, "cnet.", "batch")
            net, inp = torch.tanh(c[:, :HDIM]), torch.relu(c[:, HDIM:])
            h, w = image1.shape[2] // 8, image1.shape[3] // 8
            ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
            coords0 = torch.stack([xs, ys], 0).float()[None].repeat(n, 1, 1, 1)
            coords1 = coords0.clone()
            mask = None
            for it in range(iters):
                corr = self.lookup(pyr, coords1)
                net, mask, delta = self.update(net, inp, corr, coords1 - coords0, want_mask=(it == iters - 1))
                coords1 = coords1 + delta
            return coords1 - coords0, self.upsample(coords1 - coords0, mask)

    def flows_bi(self, frames, iters=20):
        """RAFT_bi.forward: frames [t,3,H,W] -> forward flows (i -> i+1) and backward flows (i+1 -> i), each [t-1,2,H,W]."""
        a, b = frames[:-1], frames[1:]
        return self.forward(a, b, iters)[1], self.forward(b, a, iters)[1]
