"""Restatement of the LaMa plugin (oracle; TEST INFRASTRUCTURE ONLY, see __init__.py).

Wrapper: backend/inpaint/lama_inpaint.py (LamaInpaint.inpaint :17-28, ._inpaint_batch :30-66, .__call__ :68-114) and
backend/inpaint/utils/lama_util.py (get_image :12-29, pad_img_to_modulo :52-60, prepare_img_and_mask :63-80).
PINNED: tests/golden/wrappers.npz holds what the reference's own LamaInpaint produced (executed by
oracle/make_golden_wrappers.py around the stand-in module below).

Network: the reference only ever calls an opaque TorchScript blob (`torch.jit.load('big-lama.pt')`, lama_inpaint.py:13), which is
a missing blob in the mount, and ships no source for it.  `BigLamaNet` restates the published big-LaMa
generator -- advimman/lama (the URL lama_util.py:11 cites), saicinpainting/training/modules/ffc.py `FFCResNetGenerator` with
configs/training/generator/ffc_resnet_075.yaml, wrapped as bin/export_jit.py / DefaultInpaintingTrainingModule.forward do:
    masked = image * (1 - mask);  predicted = generator(cat[masked, mask]);  out = mask * predicted + (1 - mask) * image
PARITY UNPINNED for the network (no blob, no source in the tree): key names and shapes follow the published module tree.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .sttn_auto import get_inpaint_area_by_mask


# ---- lama_util.py ---------------------------------------------------------------------------------------------
def get_image(image):
    """lama_util.py:12-29: HWC (or HW) array -> CHW float32 / 255."""
    img = np.array(image) if not isinstance(image, np.ndarray) else image.copy()
    if img.ndim == 3:
        img = np.transpose(img, (2, 0, 1))
    elif img.ndim == 2:
        img = img[np.newaxis, ...]
    assert img.ndim == 3
    return img.astype(np.float32) / 255


def ceil_modulo(x, mod):
    return x if x % mod == 0 else (x // mod + 1) * mod


def pad_img_to_modulo(img, mod):
    """lama_util.py:52-60: bottom / right padding, numpy 'symmetric' (edge pixel repeated)."""
    _, h, w = img.shape
    return np.pad(img, ((0, 0), (0, ceil_modulo(h, mod) - h), (0, ceil_modulo(w, mod) - w)), mode="symmetric")


def prepare_img_and_mask(image, mask, pad_out_to_modulo=8):
    """lama_util.py:63-80 without the (unused) scale_factor: -> image f32 [1,3,h,w], mask int64 {0,1} [1,1,h,w]."""
    out_image = pad_img_to_modulo(get_image(image), pad_out_to_modulo)
    out_mask = pad_img_to_modulo(get_image(mask), pad_out_to_modulo)
    out_image = torch.from_numpy(out_image).unsqueeze(0)
    out_mask = (torch.from_numpy(out_mask).unsqueeze(0) > 0) * 1
    return out_image, out_mask


# ---- lama_inpaint.py ------------------------------------------------------------------------------------------
class LamaOracle:
    def __init__(self, model):
        self.model = model             # callable(image f32 [B,3,h,w], mask int [B,1,h,w]) -> f32 [B,3,h,w]

    def inpaint(self, image, mask):
        """:17-28 (single image; the frames are BGR and go to the network as they are)."""
        orig_height, orig_width = np.array(image).shape[:2]
        image, mask = prepare_img_and_mask(image, mask)
        with torch.no_grad():
            inpainted = self.model(image, mask)
        cur_res = inpainted[0].permute(1, 2, 0).numpy()
        cur_res = np.clip(cur_res * 255, 0, 255).astype("uint8")
        return cur_res[:orig_height, :orig_width]

    def _inpaint_batch(self, images, masks):
        """:30-66: mini-batches of 4; a list of exactly one image goes through inpaint()."""
        if len(images) == 1:
            return [self.inpaint(images[0], masks[0])]
        orig_height, orig_width = images[0].shape[:2]
        results = [None] * len(images)
        for start in range(0, len(images), 4):
            end = min(start + 4, len(images))
            imgs = np.stack([pad_img_to_modulo(get_image(images[i]), 8) for i in range(start, end)])
            msks = np.stack([pad_img_to_modulo(get_image(masks[i]), 8) for i in range(start, end)])
            img_tensor = torch.from_numpy(imgs)
            mask_tensor = (torch.from_numpy(msks) > 0) * 1
            with torch.no_grad():
                out = self.model(img_tensor, mask_tensor).permute(0, 2, 3, 1).numpy()
            out = np.clip(out * 255, 0, 255).astype("uint8")
            for i in range(end - start):
                results[start + i] = out[i][:orig_height, :orig_width]
        return results

    def __call__(self, input_frames, input_mask):
        """:68-114: native-resolution strips of height int(W*3/16), whole strip overwritten."""
        mask = input_mask[:, :, None]
        H_ori, W_ori = mask.shape[:2]
        split_h = int(W_ori * 3 / 16)
        inpaint_area = get_inpaint_area_by_mask(W_ori, H_ori, split_h, mask)
        frames_hr = [f.copy() for f in input_frames]
        comps = {}
        for k, area in enumerate(inpaint_area):
            comps[k] = self._inpaint_batch([f[area[0]:area[1], :, :] for f in frames_hr],
                                           [mask[area[0]:area[1], :, :] for _ in frames_hr])
        if inpaint_area:
            for j, frame in enumerate(frames_hr):
                for k, area in enumerate(inpaint_area):
                    frame[area[0]:area[1], :, :] = comps[k][j]
        return frames_hr


class StandInLama(torch.nn.Module):
    """Small deterministic module with big-lama.pt's call contract (image, mask) -> inpainted; used to pin the WRAPPER."""

    def __init__(self, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w = torch.nn.Parameter(torch.randn(3, 4, 3, 3, generator=g) * 0.4, requires_grad=False)
        self.b = torch.nn.Parameter(torch.randn(3, generator=g) * 0.1, requires_grad=False)

    def forward(self, image, mask):
        masked = image * (1 - mask)
        x = torch.cat([masked, mask.to(image.dtype)], dim=1)
        y = torch.sigmoid(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), self.w, self.b))
        return mask * y + (1 - mask) * image


# ---- the published big-LaMa generator ---------------------------------------------------------------------------------
class BigLamaNet:
    """forward(image f32 [B,3,h,w] in [0,1], mask {0,1} [B,1,h,w]) -> inpainted f32 [B,3,h,w]; h, w multiples of 8.
    State-dict keys: `model.N...` of FFCResNetGenerator (an optional `generator.` prefix is dropped)."""

    def __init__(self, state_dict, n_blocks=18):
        self.sd = {(k[10:] if k.startswith("generator.") else k): torch.as_tensor(np.asarray(v)).float() for k, v in state_dict.items()}
        self.n_blocks = n_blocks

    def _bn(self, x, p):
        sd = self.sd
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)

    def _rconv(self, x, key, stride=1, pad=1):
        """nn.Conv2d(..., padding=pad, padding_mode='reflect', bias=False) of FFC (ffc.py FFC.__init__)"""
        return F.conv2d(F.pad(x, (pad,) * 4, mode="reflect") if pad else x, self.sd[key + ".weight"], None, stride)

    def _fourier_unit(self, x, p):
        """FourierUnit.forward: rfftn(norm='ortho') -> 1x1 conv on stacked (re, im) channels + BN + ReLU -> irfftn"""
        b, c, h, w = x.shape
        f = torch.fft.rfftn(x, dim=(-2, -1), norm="ortho")
        f = torch.stack((f.real, f.imag), dim=-1).permute(0, 1, 4, 2, 3).contiguous().view(b, 2 * c, h, w // 2 + 1)
        f = F.relu(self._bn(F.conv2d(f, self.sd[p + ".conv_layer.weight"]), p + ".bn"))
        f = f.view(b, c, 2, h, w // 2 + 1).permute(0, 1, 3, 4, 2).contiguous()
        return torch.fft.irfftn(torch.complex(f[..., 0], f[..., 1]), s=(h, w), dim=(-2, -1), norm="ortho")

    def _spectral(self, x, p):
        """SpectralTransform.forward, stride 1, enable_lfu False"""
        x = F.relu(self._bn(F.conv2d(x, self.sd[p + ".conv1.0.weight"]), p + ".conv1.1"))
        return F.conv2d(x + self._fourier_unit(x, p + ".fu"), self.sd[p + ".conv2.weight"])

    def _ffc_block(self, x_l, x_g, p):
        """FFC_BN_ACT with ratio_gin = ratio_gout = 0.75, 3x3, reflect padding 1"""
        out_l = self._rconv(x_l, p + ".ffc.convl2l") + self._rconv(x_g, p + ".ffc.convg2l")
        out_g = self._rconv(x_l, p + ".ffc.convl2g") + self._spectral(x_g, p + ".ffc.convg2g")
        return F.relu(self._bn(out_l, p + ".bn_l")), F.relu(self._bn(out_g, p + ".bn_g"))

    def generator(self, x):
        sd = self.sd
        x = F.relu(self._bn(self._rconv(x, "model.1.ffc.convl2l", 1, 3), "model.1.bn_l"))
        x = F.relu(self._bn(self._rconv(x, "model.2.ffc.convl2l", 2, 1), "model.2.bn_l"))
        x = F.relu(self._bn(self._rconv(x, "model.3.ffc.convl2l", 2, 1), "model.3.bn_l"))
        x_l = F.relu(self._bn(self._rconv(x, "model.4.ffc.convl2l", 2, 1), "model.4.bn_l"))
        x_g = F.relu(self._bn(self._rconv(x, "model.4.ffc.convl2g", 2, 1), "model.4.bn_g"))
        for i in range(self.n_blocks):
            p = f"model.{5 + i}"
            y_l, y_g = self._ffc_block(x_l, x_g, p + ".conv1")
            y_l, y_g = self._ffc_block(y_l, y_g, p + ".conv2")
            x_l, x_g = x_l + y_l, x_g + y_g
        x = torch.cat([x_l, x_g], dim=1)
        base = 5 + self.n_blocks + 1
        for j in range(3):
            k = f"model.{base + 3 * j}"
            x = F.conv_transpose2d(x, sd[k + ".weight"], sd[k + ".bias"], stride=2, padding=1, output_padding=1)
            x = F.relu(self._bn(x, f"model.{base + 3 * j + 1}"))
        k = f"model.{base + 10}"
        return torch.sigmoid(F.conv2d(F.pad(x, (3,) * 4, mode="reflect"), sd[k + ".weight"], sd[k + ".bias"]))

    def __call__(self, image, mask):
        with torch.no_grad():
            masked = image * (1 - mask)
            pred = self.generator(torch.cat([masked, mask.to(image.dtype)], dim=1))
            return mask * pred + (1 - mask) * image
