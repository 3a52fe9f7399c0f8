"""CPU restatement of torchvision.ops.deform_conv2d (test infrastructure only).

torchvision 0.22.0 (the reference's pin, README_en.md:166 / docker image) is absent from this image and from the
reference mount, so the operator is restated from its published definition (torchvision/ops/deform_conv.py docstring
and torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp, `bilinear_interpolate` / `deformable_im2col_kernel`):

  out[n,co,y,x] = bias[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] * mask[n, g*K + k, y, x]
                  * bilinear(in[n,ci], y*s - p + ky*d + offset[n, g*2K + 2k, y, x], x*s - p + kx*d + offset[n, g*2K + 2k + 1, y, x])

with k = ky*kw + kx, K = kh*kw, g = ci // (Cin / offset_groups), and a bilinear sample whose four corners count as zero
outside the image.  Call sites in the reference: recurrent_flow_completion.py:44-46, propainter.py:70-72 (weight groups
= 1, stride 1, padding 1, dilation 1, 16 offset groups).  Parity of everything that calls it is pinned to THIS
restatement, not to torchvision's binary: "parity unpinned" for the operator itself.
"""
import torch


def deform_conv2d(x, offset, weight, bias=None, stride=1, padding=1, dilation=1, mask=None):
    stride = stride[0] if isinstance(stride, (tuple, list)) else stride
    padding = padding[0] if isinstance(padding, (tuple, list)) else padding
    dilation = dilation[0] if isinstance(dilation, (tuple, list)) else dilation
    n, cin, H, W = x.shape
    cout, cin_w, kh, kw = weight.shape
    assert cin_w == cin, "weight groups = 1 on this path"
    K = kh * kw
    G = offset.shape[1] // (2 * K)
    oh = (H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    ow = (W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    assert offset.shape[2:] == (oh, ow)
    cpg = cin // G
    ys = torch.arange(oh, dtype=x.dtype).view(1, 1, oh, 1) * stride - padding
    xs = torch.arange(ow, dtype=x.dtype).view(1, 1, 1, ow) * stride - padding
    off = offset.view(n, G, K, 2, oh, ow)
    cols = x.new_zeros(n, cin, K, oh, ow)
    xg = x.view(n, G, cpg, H * W)
    for k in range(K):
        ky, kx = divmod(k, kw)
        py = ys + ky * dilation + off[:, :, k, 0]                    # [n,G,oh,ow]
        px = xs + kx * dilation + off[:, :, k, 1]
        y0, x0 = torch.floor(py), torch.floor(px)
        ly, lx = py - y0, px - x0
        y0, x0 = y0.long(), x0.long()
        acc = x.new_zeros(n, G, cpg, oh, ow)
        for dy, dx, wgt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx), (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
            yy, xx = y0 + dy, x0 + dx
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).view(n, G, 1, oh * ow).expand(n, G, cpg, oh * ow)
            v = torch.gather(xg, 3, idx).view(n, G, cpg, oh, ow)
            acc = acc + v * (wgt * ok.to(x.dtype)).unsqueeze(2)
        if mask is not None:
            acc = acc * mask.view(n, G, K, oh, ow)[:, :, k].unsqueeze(2)
        cols[:, :, k] = acc.view(n, cin, oh, ow)
    out = torch.einsum("ok,nkp->nop", weight.reshape(cout, cin * K), cols.view(n, cin * K, oh * ow)).view(n, cout, oh, ow)
    if bias is not None:
        out = out + bias.view(1, cout, 1, 1)
    return out
