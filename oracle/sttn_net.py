"""fp32 torch-CPU restatement of the reference STTN generators (oracle; see __init__.py).

Follows backend/inpaint/sttn/auto_sttn.py (sttn-auto) and backend/inpaint/sttn/network_sttn.py
(sttn-det; same parameter structure, different patch table, and an attention mask that the
reference computes but never applies -- network_sttn.py:149 drops the masked_fill result).
Written functionally over a state_dict so that it needs nothing from /root/reference at run time.
"""
import math

import torch
import torch.nn.functional as F

PATCHSIZE = {
    "auto": [(80, 15), (32, 6), (10, 5), (5, 3)],     # auto_sttn.py:69  (width, height)
    "det": [(108, 60), (36, 20), (18, 10), (9, 5)],   # network_sttn.py:69
}
MODEL_SIZE = {"auto": (640, 120), "det": (432, 240)}   # (w, h): sttn_auto_inpaint.py:39 / sttn_det_inpaint.py
CHANNEL = 256
STACK_NUM = 8


def _lrelu(x):
    return F.leaky_relu(x, 0.2)


class SttnNet:
    """InpaintGenerator restated (auto_sttn.py:64-115 / network_sttn.py:64-121)."""

    def __init__(self, state_dict, variant="auto"):
        self.variant = variant
        self.patchsize = PATCHSIZE[variant]
        self.w = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in state_dict.items()}

    # auto_sttn.py:75-84
    def encoder(self, x):
        w = self.w
        x = _lrelu(F.conv2d(x, w["encoder.0.weight"], w["encoder.0.bias"], stride=2, padding=1))
        x = _lrelu(F.conv2d(x, w["encoder.2.weight"], w["encoder.2.bias"], stride=1, padding=1))
        x = _lrelu(F.conv2d(x, w["encoder.4.weight"], w["encoder.4.bias"], stride=2, padding=1))
        x = _lrelu(F.conv2d(x, w["encoder.6.weight"], w["encoder.6.bias"], stride=1, padding=1))
        return x

    # auto_sttn.py:118-127 (deconv) and :87-95
    @staticmethod
    def _deconv(x, weight, bias):
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        return F.conv2d(x, weight, bias, stride=1, padding=1)

    def decoder(self, x):
        w = self.w
        x = _lrelu(self._deconv(x, w["decoder.0.conv.weight"], w["decoder.0.conv.bias"]))
        x = _lrelu(F.conv2d(x, w["decoder.2.weight"], w["decoder.2.bias"], stride=1, padding=1))
        x = _lrelu(self._deconv(x, w["decoder.4.conv.weight"], w["decoder.4.conv.bias"]))
        x = F.conv2d(x, w["decoder.6.weight"], w["decoder.6.bias"], stride=1, padding=1)
        return x

    # auto_sttn.py:167-206 + Attention :140-145
    def _attention(self, x, p, b=1):
        w = self.w
        bt, c, h, wd = x.shape
        t = bt // b
        d_k = c // len(self.patchsize)
        _query = F.conv2d(x, w[p + "query_embedding.weight"], w[p + "query_embedding.bias"])
        _key = F.conv2d(x, w[p + "key_embedding.weight"], w[p + "key_embedding.bias"])
        _value = F.conv2d(x, w[p + "value_embedding.weight"], w[p + "value_embedding.bias"])
        output = []
        n = len(self.patchsize)
        for (width, height), query, key, value in zip(self.patchsize, torch.chunk(_query, n, dim=1),
                                                      torch.chunk(_key, n, dim=1), torch.chunk(_value, n, dim=1)):
            out_w, out_h = wd // width, h // height

            def split(z):
                z = z.reshape(b, t, d_k, out_h, height, out_w, width)
                return z.permute(0, 1, 3, 5, 2, 4, 6).contiguous().view(b, t * out_h * out_w, d_k * height * width)

            q, k, v = split(query), split(key), split(value)
            scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(q.size(-1))
            p_attn = F.softmax(scores, dim=-1)
            y = torch.matmul(p_attn, v)
            y = y.view(b, t, out_h, out_w, d_k, height, width)
            y = y.permute(0, 1, 4, 2, 5, 3, 6).contiguous().view(bt, d_k, h, wd)
            output.append(y)
        output = torch.cat(output, 1)
        return _lrelu(F.conv2d(output, w[p + "output_linear.0.weight"], w[p + "output_linear.0.bias"], padding=1))

    # auto_sttn.py:210-239
    def _block(self, x, i):
        w = self.w
        p = f"transformer.{i}."
        x = x + self._attention(x, p + "attention.")
        y = _lrelu(F.conv2d(x, w[p + "feed_forward.conv.0.weight"], w[p + "feed_forward.conv.0.bias"],
                            padding=2, dilation=2))
        y = _lrelu(F.conv2d(y, w[p + "feed_forward.conv.2.weight"], w[p + "feed_forward.conv.2.bias"], padding=1))
        return x + y

    # auto_sttn.py:111-115
    def infer(self, feat):
        x = feat
        for i in range(STACK_NUM):
            x = self._block(x, i)
        return x


def to_torch_module_state(state_dict):
    return {k: torch.as_tensor(v) for k, v in state_dict.items()}
