"""Deterministic synthetic STTN weights with the reference's state_dict keys and shapes.

The shipped checkpoints (backend/models/sttn-auto/infer_model.pth, sttn-det/sttn.pth) are
missing blobs, and the reference's own init (auto_sttn.py:24-61, normal(0, 0.02), zero bias)
makes a degenerate network whose output is a constant grey image (features shrink ~300x
through the encoder), so parity against it would not see real errors.  These weights are
variance-preserving (std = gain / sqrt(fan_in) per layer, small non-zero biases) so that the
output spans the whole uint8 range and the attention is neither uniform nor one-hot; they are
loaded into the reference module with ``load_state_dict(strict=True)`` by make_golden.py,
exactly like a real checkpoint (sttn_auto_inpaint.py:34).
"""
import numpy as np

_GAINS = dict(enc=1.8, qk=0.8, v=1.0, tr=0.35, dec=1.3, last=0.7)
_BIAS_STD = 0.02


def state_dict_spec(variant="auto"):
    """(key, shape) in the reference's state_dict order (auto_sttn.py:64-95 / network_sttn.py:64-95)."""
    spec = []
    c = 256
    for i in range(8):
        p = f"transformer.{i}."
        for name, k in (("attention.query_embedding", 1), ("attention.value_embedding", 1),
                        ("attention.key_embedding", 1), ("attention.output_linear.0", 3),
                        ("feed_forward.conv.0", 3), ("feed_forward.conv.2", 3)):
            spec.append((p + name + ".weight", (c, c, k, k)))
            spec.append((p + name + ".bias", (c,)))
    for name, co, ci in (("encoder.0", 64, 3), ("encoder.2", 64, 64), ("encoder.4", 128, 64),
                         ("encoder.6", 256, 128), ("decoder.0.conv", 128, 256), ("decoder.2", 64, 128),
                         ("decoder.4.conv", 64, 64), ("decoder.6", 3, 64)):
        spec.append((name + ".weight", (co, ci, 3, 3)))
        spec.append((name + ".bias", (co,)))
    return spec


def _gain_for(key):
    if key.startswith("encoder"):
        return _GAINS["enc"]
    if "query" in key or "key_embedding" in key:
        return _GAINS["qk"]
    if "value" in key:
        return _GAINS["v"]
    if key.startswith("transformer"):
        return _GAINS["tr"]
    if key.startswith("decoder.6"):
        return _GAINS["last"]
    return _GAINS["dec"]


def make_state_dict(seed=0, variant="auto"):
    """dict key -> float32 ndarray; numpy PCG64 so it is identical on every machine."""
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape in state_dict_spec(variant):
        if key.endswith("weight"):
            fan_in = int(np.prod(shape[1:]))
            sd[key] = (rng.standard_normal(shape).astype(np.float32)
                       * np.float32(_gain_for(key) / np.sqrt(fan_in)))
        else:
            sd[key] = rng.standard_normal(shape).astype(np.float32) * np.float32(_BIAS_STD)
    return sd
