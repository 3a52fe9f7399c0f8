"""Generate tests/golden/* from the REFERENCE implementation (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

Imports the reference's own nn.Modules from /root/reference (read-only; import stubs for cv2 /
torchvision only satisfy unrelated top-level imports: backend/inpaint/utils/__init__.py star-imports
a cv2-using helper, network_sttn.py:8 imports unused torchvision.models), loads the synthetic
weights of oracle/weights.py with load_state_dict(strict=True) -- which also pins the key names
and shapes -- and records outputs on seeded inputs.  /root/reference does not exist on the GPU
box: tests only read the committed fixtures.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    for n in ("cv2", "torchvision", "torchvision.models"):
        sys.modules.setdefault(n, types.ModuleType(n))
    cfg = types.ModuleType("backend.config")          # qfluentwidgets-free stand-in for inpaint_tools
    cfg.config = types.SimpleNamespace()
    sys.path.insert(0, REF)
    from backend.inpaint.sttn import auto_sttn, network_sttn
    sys.modules["backend.config"] = cfg
    from backend.tools import inpaint_tools
    return auto_sttn, network_sttn, inpaint_tools


def _sample(t, n=4096, seed=123):
    flat = t.detach().reshape(-1).numpy()
    idx = np.random.default_rng(seed).integers(0, flat.size, size=n)
    return idx.astype(np.int64), flat[idx].astype(np.float32)


def main():
    from vsr_amd.synth import make_state_dict

    os.makedirs(OUT, exist_ok=True)
    auto_sttn, network_sttn, inpaint_tools = _import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))

    # ---- sttn-auto generator: encoder -> infer -> decoder -> tanh on 3 frames of 120x640 ----
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(0, "auto").items()}
    net = auto_sttn.InpaintGenerator(init_weights=False).eval()
    net.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(7)
    frames = rng.integers(0, 256, size=(3, 120, 640, 3), dtype=np.uint8)      # BGR
    x = torch.from_numpy(np.ascontiguousarray(frames[..., ::-1])).permute(0, 3, 1, 2).float().div(255) * 2 - 1
    with torch.no_grad():
        feat = net.encoder(x)
        pred = net.infer(feat)
        out = torch.tanh(net.decoder(pred[:2]))
    fi, fv = _sample(feat)
    pi, pv = _sample(pred)
    np.savez_compressed(
        os.path.join(OUT, "sttn_auto_net.npz"), frames_seed=7,
        feat_idx=fi, feat_val=fv, feat_sum=np.float64(feat.double().sum()), feat_sq=np.float64((feat.double() ** 2).sum()),
        pred_idx=pi, pred_val=pv, pred_sum=np.float64(pred.double().sum()), pred_sq=np.float64((pred.double() ** 2).sum()),
        out_sub=out[:, :, ::2, ::4].numpy().astype(np.float32),
        out_sum=np.float64(out.double().sum()), out_sq=np.float64((out.double() ** 2).sum()))

    # ---- sttn-det generator (mask input must have no effect: network_sttn.py:149) ----
    sd = {k: torch.from_numpy(v) for k, v in make_state_dict(1, "det").items()}
    net = network_sttn.InpaintGenerator(init_weights=False).eval()
    net.load_state_dict(sd, strict=True)
    rng = np.random.default_rng(8)
    x = torch.from_numpy(rng.random((2, 3, 240, 432), dtype=np.float32) * 2 - 1)
    masks = torch.from_numpy((rng.random((2, 1, 240, 432)) > 0.7).astype(np.float32))
    with torch.no_grad():
        feat = net.encoder(x)
        pred = net.infer(feat, masks)
        out = torch.tanh(net.decoder(pred[:1]))
    pi, pv = _sample(pred)
    np.savez_compressed(
        os.path.join(OUT, "sttn_det_net.npz"), x_seed=8,
        pred_idx=pi, pred_val=pv, pred_sum=np.float64(pred.double().sum()), pred_sq=np.float64((pred.double() ** 2).sum()),
        out_sub=out[:, :, ::4, ::4].numpy().astype(np.float32),
        out_sum=np.float64(out.double().sum()), out_sq=np.float64((out.double() ** 2).sum()))

    # ---- batch_generator (tools/inpaint_tools.py:7-29), executed from the reference ----
    cases = [(1200, 50), (300, 50), (600, 50), (1200, 70), (49, 50), (50, 50), (51, 50), (75, 50), (1, 50), (0, 50),
             (10, 3), (7, 1), (99, 10), (100, 10), (101, 10)]
    res = {f"{n},{m}": [len(b) for b in inpaint_tools.batch_generator(list(range(n)), m)] for n, m in cases}
    with open(os.path.join(OUT, "batch_generator.json"), "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
