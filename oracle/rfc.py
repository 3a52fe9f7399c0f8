"""CPU restatement of the reference's RecurrentFlowCompleteNet (test infrastructure only) -- SURVEY.md 8(a) row a15.

Follows backend/inpaint/video/model/recurrent_flow_completion.py:
  RecurrentFlowCompleteNet.forward :275-311, forward_bidirect_flow :313-339, combine_flow :341-348
  P3DBlock :152-173, mid_dilation :229-236, decoders / deconv :129-149,241-259
  BidirectionalPropagation.forward :69-126, SecondOrderDeformableAlignment.forward :31-46
torchvision.ops.deform_conv2d is restated in oracle/deform_conv.py (absent dependency).  Pinned by
oracle/make_golden.py against the reference module run with that same operator restatement (tests/golden/rfc.npz).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .deform_conv import deform_conv2d


class RfcOracle:
    def __init__(self, state_dict):
        self.sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in state_dict.items()}

    def c2(self, x, name, padding=1, dilation=1):
        return F.conv2d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], padding=padding, dilation=dilation)

    def c3(self, x, name, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1)):
        return F.conv3d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], stride=stride, padding=padding, dilation=dilation)

    def p3d(self, x, name, stride, residual=False):
        y = F.leaky_relu(self.c3(x, name + ".conv1.0", (1, stride, stride), (0, 1, 1)), 0.2)
        y = self.c3(y, name + ".conv2.0", (1, 1, 1), (2, 0, 0), (2, 1, 1))
        return x + y if residual else y

    def deform_align(self, mod, x, cond):
        p = f"feat_prop_module.deform_align.{mod}"
        o = cond
        for i in (0, 2, 4):
            o = F.leaky_relu(self.c2(o, f"{p}.conv_offset.{i}"), 0.1)
        o = self.c2(o, f"{p}.conv_offset.6")
        o1, o2, m = torch.chunk(o, 3, dim=1)
        offset = 5.0 * torch.tanh(torch.cat((o1, o2), dim=1))          # max_residue_magnitude = 5
        return deform_conv2d(x, offset, self.sd[p + ".weight"], self.sd[p + ".bias"], 1, 1, 1, torch.sigmoid(m))

    def propagate(self, x):
        """BidirectionalPropagation.forward: x [t,c,h,w] (batch 1) -> [t,c,h,w]"""
        t = x.shape[0]
        feats = {"spatial": [x[i:i + 1] for i in range(t)]}
        for mod in ("backward_", "forward_"):
            feats[mod] = []
            order = list(range(t))[::-1] if mod == "backward_" else list(range(t))
            prop = torch.zeros_like(x[:1])
            for i, idx in enumerate(order):
                cur = feats["spatial"][idx]
                if i > 0:
                    n2 = feats[mod][-2] if i > 1 else torch.zeros_like(prop)
                    cond = torch.cat([prop, cur, n2], 1)
                    prop = self.deform_align(mod, torch.cat([prop, n2], 1), cond)
                parts = [cur] + [feats[k][idx] for k in feats if k not in ("spatial", mod)] + [prop]
                b = f"feat_prop_module.backbone.{mod}"
                y = self.c2(F.leaky_relu(self.c2(torch.cat(parts, 1), b + ".0"), 0.1), b + ".2")
                prop = prop + y
                feats[mod].append(prop)
            if mod == "backward_":
                feats[mod] = feats[mod][::-1]
        outs = [self.c2(torch.cat([feats["backward_"][i], feats["forward_"][i]], 1), "feat_prop_module.fusion", padding=0) for i in range(t)]
        return torch.cat(outs, 0) + x

    def deconv(self, x, name):
        return self.c2(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True), name + ".conv")

    def forward(self, masked_flows, masks):
        """masked_flows [t,2,h,w], masks [t,1,h,w] (batch 1) -> completed flows [t,2,h,w]"""
        with torch.no_grad():
            t, _, h, w = masked_flows.shape
            inp = torch.cat((masked_flows, masks), 1).permute(1, 0, 2, 3)[None]                      # [1,3,t,h,w]
            x = F.conv3d(F.pad(inp, (2, 2, 2, 2, 0, 0), mode="replicate"), self.sd["downsample.0.weight"], self.sd["downsample.0.bias"],
                         stride=(1, 2, 2))
            x = F.leaky_relu(x, 0.2)
            e1 = F.leaky_relu(self.p3d(F.leaky_relu(self.p3d(x, "encoder1.0", 1), 0.2), "encoder1.2", 2), 0.2)
            e2 = F.leaky_relu(self.p3d(F.leaky_relu(self.p3d(e1, "encoder2.0", 1), 0.2), "encoder2.2", 2), 0.2)
            mid = e2
            for i, d in ((0, 3), (2, 2), (4, 1)):
                mid = F.leaky_relu(self.c3(mid, f"mid_dilation.{i}", (1, 1, 1), (0, d, d), (1, d, d)), 0.2)
            prop = self.propagate(mid[0].permute(1, 0, 2, 3))                                        # [t,128,h/8,w/8]
            e1f = e1[0].permute(1, 0, 2, 3)
            d2 = F.leaky_relu(self.deconv(F.leaky_relu(self.c2(prop, "decoder2.0"), 0.2), "decoder2.2"), 0.2) + e1f
            d1 = F.leaky_relu(self.deconv(F.leaky_relu(self.c2(d2, "decoder1.0"), 0.2), "decoder1.2"), 0.2)
            return self.deconv(F.leaky_relu(self.c2(d1, "upsample.0"), 0.2), "upsample.2")

    def complete_bi(self, flows_f, flows_b, masks):
        """forward_bidirect_flow + combine_flow: flows [t-1,2,h,w] each, masks [t,1,h,w] in {0,1} -> completed (fwd, bwd)"""
        mf, mb = masks[:-1], masks[1:]
        in_f, in_b = flows_f * (1 - mf), flows_b * (1 - mb)
        pf = self.forward(in_f, mf)
        pb = torch.flip(self.forward(torch.flip(in_b, dims=[0]), torch.flip(mb, dims=[0])), dims=[0])
        return pf * mf + in_f * (1 - mf), pb * mb + in_b * (1 - mb), pf, pb
