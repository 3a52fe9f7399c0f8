"""Seeded synthetic inputs for parity tests and bench.py: clips and
stand-in STTN checkpoints (the shipped .pth files are missing blobs in the reference mount).

make_clip: a smooth low-frequency background that translates a few pixels per frame (so temporal
attention has signal), plus white "subtitle" glyph blocks inside the box.  uint8 BGR frames.

make_state_dict: The shipped checkpoints (backend/models/sttn-auto/infer_model.pth, sttn-det/sttn.pth) are
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

# Weight-statistics profiles (round 6).  Every parity number of rounds 1-5 was taken on ONE benign draw per network:
# Gaussian, variance-preserving.  Trained checkpoints are not like that, so each stand-in checkpoint can also be drawn
#   "peaked"    query / key gain x4: attention rows close to one-hot (what a trained attention looks like);
#   "heavy"     Student-t (nu = 3) weights of the same variance -- a few weights many sigmas out -- and the first layer scaled up by
#               HEAVY_SCALE with the last layer scaled down by the same factor: the activations in between sit a few times below the
#               fp16 limit (65504), where the fp16-operand modes either hold their accuracy or must trip the range guard;
#   "undamped"  RAFT only: the flow head's last conv at the gain of every other conv -- flow updates of tens of pixels per iteration.
# The same seeds as the benign draw; oracle/make_golden_sweep.py loads every profile into the reference modules (strict=True).
PROFILES = ("benign", "peaked", "heavy", "undamped")
HEAVY_SCALE = dict(sttn=600.0, propainter=30.0, propainter_ffn=1000.0, rfc=900.0, raft=1.0)
# (ProPainter: the feature propagation gates its deformable offsets with tanh / sigmoid of conv outputs -- scaled by hundreds every gate
#  saturates and the REFERENCE module in fp32 lands 0.14-0.32 of the tanh range away from its own float64 run, no fp32 implementation can
#  be compared on that.  So the encoder is scaled by 30 only (fp32-vs-float64 gap 3e-6) and the large activations are put where the
#  network is smooth: every block's fc1 x 1000, fc2 / 1000 -- the FFN hidden tensor, the largest fp16 operand of the generator.)


def _draw(rng, shape, profile):
    """unit-variance weights of a profile (float32)"""
    if profile == "heavy":
        return (rng.standard_t(3, shape) / np.sqrt(3.0)).astype(np.float32)
    return rng.standard_normal(shape).astype(np.float32)


def _check_profile(profile):
    if profile not in PROFILES:
        raise ValueError(f"weight profile {profile!r}: expected one of {PROFILES}")


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


def make_state_dict(seed=0, variant="auto", profile="benign"):
    """dict key -> float32 ndarray; numpy PCG64 so it is identical on every machine.  profile: PROFILES above."""
    _check_profile(profile)
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape in state_dict_spec(variant):
        if key.endswith("weight"):
            fan_in = int(np.prod(shape[1:]))
            gain = _gain_for(key)
            if profile == "peaked" and ("query" in key or "key_embedding" in key):
                gain *= 4.0
            if profile == "heavy" and key.startswith("encoder.0"):
                gain *= HEAVY_SCALE["sttn"]
            if profile == "heavy" and key.startswith("decoder.6"):
                gain /= HEAVY_SCALE["sttn"]
            sd[key] = _draw(rng, shape, profile) * np.float32(gain / np.sqrt(fan_in))
        else:
            sd[key] = rng.standard_normal(shape).astype(np.float32) * np.float32(_BIAS_STD)
    return sd


def make_clip(n, H, W, box, seed=0, glyph_frames=None):
    """box = (ymin, ymax, xmin, xmax) of the subtitle area (CLI order, args_handler.py:19).  glyph_frames: per-frame booleans --
    frames without the glyph blocks are the background alone (what make_clip gives for an empty box with the same seed)."""
    rng = np.random.default_rng(seed)
    gh, gw = H // 40 + 3, W // 40 + 3
    base = rng.random((gh, gw, 3)).astype(np.float32)
    ys = np.linspace(0, gh - 2, H + 64).astype(np.float32)
    xs = np.linspace(0, gw - 2, W + 64).astype(np.float32)
    y0 = np.floor(ys).astype(int)
    x0 = np.floor(xs).astype(int)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    big = ((1 - fy) * (1 - fx) * base[y0][:, x0] + (1 - fy) * fx * base[y0][:, x0 + 1]
           + fy * (1 - fx) * base[y0 + 1][:, x0] + fy * fx * base[y0 + 1][:, x0 + 1])
    big = (big * 200 + 25).astype(np.float32)
    ymin, ymax, xmin, xmax = box
    frames = np.empty((n, H, W, 3), dtype=np.uint8)
    for i in range(n):
        dy, dx = (i * 2) % 64, (i * 3) % 64
        img = big[dy:dy + H, dx:dx + W] + rng.normal(0, 2.0, (H, W, 3)).astype(np.float32)
        img = np.clip(img, 0, 255).astype(np.uint8)
        # glyph blocks: white rectangles with dark outline, text changes every 24 frames
        grng = np.random.default_rng(seed * 1000 + i // 24)
        gh_px = max((ymax - ymin) // 2, 4)
        gy = ymin + (ymax - ymin - gh_px) // 2
        x = xmin + 8
        while (glyph_frames is None or glyph_frames[i]) and x + gh_px < xmax - 8:
            wpx = int(grng.integers(gh_px // 2, gh_px + 1))
            if grng.random() < 0.8:
                img[gy:gy + gh_px, x:x + wpx] = 16
                img[gy + 2:gy + gh_px - 2, x + 2:x + wpx - 2] = 250
            x += wpx + max(gh_px // 4, 2)
        frames[i] = img
    return frames


# ------------------------------------------------------------------------------------------------
# RAFT (backend/inpaint/video/raft/raft.py:24-57, "things" checkpoint layout, args.small = False)
# ------------------------------------------------------------------------------------------------
def _raft_encoder_spec(prefix, out_dim, batch_norm):
    """BasicEncoder (raft/extractor.py:118-160).  InstanceNorm2d has no parameters; BatchNorm2d has five entries,
    and a stride-2 block's norm3 is ALSO registered as downsample.1 (extractor.py:46-47), so both keys exist."""
    spec = []

    def conv(name, co, ci, kh, kw):
        spec.append((f"{prefix}{name}.weight", (co, ci, kh, kw)))
        spec.append((f"{prefix}{name}.bias", (co,)))

    def norm(name, c):
        if batch_norm:
            for leaf, shape in (("weight", (c,)), ("bias", (c,)), ("running_mean", (c,)), ("running_var", (c,)),
                                ("num_batches_tracked", ())):
                spec.append((f"{prefix}{name}.{leaf}", shape))

    norm("norm1", 64)
    conv("conv1", 64, 3, 7, 7)
    cin = 64
    for li, (dim, stride) in enumerate(((64, 1), (96, 2), (128, 2)), start=1):
        for bi in range(2):
            p = f"layer{li}.{bi}."
            s = stride if bi == 0 else 1
            conv(p + "conv1", dim, cin, 3, 3)
            conv(p + "conv2", dim, dim, 3, 3)
            norm(p + "norm1", dim)
            norm(p + "norm2", dim)
            if s != 1:
                norm(p + "norm3", dim)
                conv(p + "downsample.0", dim, cin, 1, 1)
                norm(p + "downsample.1", dim)
            cin = dim
    conv("conv2", out_dim, 128, 1, 1)
    return spec


def raft_state_dict_spec():
    spec = _raft_encoder_spec("fnet.", 256, False) + _raft_encoder_spec("cnet.", 256, True)
    u = "update_block."
    for name, co, ci, kh, kw in (("encoder.convc1", 256, 324, 1, 1), ("encoder.convc2", 192, 256, 3, 3),
                                 ("encoder.convf1", 128, 2, 7, 7), ("encoder.convf2", 64, 128, 3, 3),
                                 ("encoder.conv", 126, 256, 3, 3),
                                 ("gru.convz1", 128, 384, 1, 5), ("gru.convr1", 128, 384, 1, 5), ("gru.convq1", 128, 384, 1, 5),
                                 ("gru.convz2", 128, 384, 5, 1), ("gru.convr2", 128, 384, 5, 1), ("gru.convq2", 128, 384, 5, 1),
                                 ("flow_head.conv1", 256, 128, 3, 3), ("flow_head.conv2", 2, 256, 3, 3),
                                 ("mask.0", 256, 128, 3, 3), ("mask.2", 576, 256, 1, 1)):
        spec.append((u + name + ".weight", (co, ci, kh, kw)))
        spec.append((u + name + ".bias", (co,)))
    return spec


def make_raft_state_dict(seed=0, profile="benign"):
    """Stand-in for weights/raft-things.pth (missing blob): He-style conv weights, non-trivial BatchNorm statistics,
    a damped flow head so that 20 GRU iterations stay in a few-pixel regime ("undamped": it is not; "heavy": Student-t weights).
    Keys as saved by the reference's checkpoint minus DataParallel's "module." prefix (flow_comp_raft.py:17-19)."""
    _check_profile(profile)
    rng = np.random.default_rng(seed + 77)
    sd = {}
    for key, shape in raft_state_dict_spec():
        leaf = key.rsplit(".", 1)[1]
        is_norm = ".norm" in key or "downsample.1" in key
        if key.endswith("norm3." + leaf):                 # alias of downsample.1: filled when that key comes
            continue
        if is_norm:
            if leaf == "weight":
                v = rng.uniform(0.6, 1.4, shape)
            elif leaf == "bias":
                v = rng.normal(0, 0.1, shape)
            elif leaf == "running_mean":
                v = rng.normal(0, 0.2, shape)
            elif leaf == "running_var":
                v = rng.uniform(0.5, 1.5, shape)
            else:
                v = np.zeros(shape)
            sd[key] = np.asarray(v, dtype=np.int64 if leaf == "num_batches_tracked" else np.float32)
            if "downsample.1" in key:
                sd[key.replace("downsample.1", "norm3")] = sd[key]
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:]))
            gain = 1.3
            if "flow_head.conv2" in key:
                gain = 1.3 if profile == "undamped" else 0.25
            elif "gru.conv" in key or "mask.2" in key:
                gain = 0.9
            sd[key] = _draw(rng, shape, profile) * np.float32(gain / np.sqrt(fan_in))
        else:
            sd[key] = rng.standard_normal(shape).astype(np.float32) * np.float32(0.05)
    # the reference's state_dict order lists norm3 before downsample.*; order does not matter for loading
    return sd


def make_flow_frames(t, H, W, seed=0):
    """t RGB uint8 frames [t,H,W,3] of a smooth texture translating by a few pixels per frame plus an
    independently moving bright block -- something an optical-flow network can lock onto."""
    rng = np.random.default_rng(seed + 4242)
    gh, gw = H // 12 + 6, W // 12 + 6
    base = rng.random((gh, gw, 3)).astype(np.float32)
    pad = 48
    ys = np.linspace(0, gh - 2, H + pad).astype(np.float32)
    xs = np.linspace(0, gw - 2, W + pad).astype(np.float32)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    big = ((1 - fy) * (1 - fx) * base[y0][:, x0] + (1 - fy) * fx * base[y0][:, x0 + 1]
           + fy * (1 - fx) * base[y0 + 1][:, x0] + fy * fx * base[y0 + 1][:, x0 + 1]) * 220 + 20
    out = np.empty((t, H, W, 3), dtype=np.uint8)
    for i in range(t):
        dy, dx = (2 * i) % pad, (3 * i) % pad
        img = big[dy:dy + H, dx:dx + W].copy()
        by, bx = H // 3 + 2 * i, W // 4 + 5 * i
        img[by:by + H // 6, bx:bx + W // 8] = 240
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


# ------------------------------------------------------------------------------------------------
# RecurrentFlowCompleteNet (backend/inpaint/video/model/recurrent_flow_completion.py:206-273)
# ------------------------------------------------------------------------------------------------
def rfc_state_dict_spec():
    spec = []

    def add(name, *shape):
        spec.append((name + ".weight", tuple(shape)))
        spec.append((name + ".bias", (shape[0],)))

    add("downsample.0", 32, 3, 1, 5, 5)
    for name, cin, cout in (("encoder1.0", 32, 32), ("encoder1.2", 32, 64), ("encoder2.0", 64, 64), ("encoder2.2", 64, 128)):
        add(name + ".conv1.0", cout, cin, 1, 3, 3)
        add(name + ".conv2.0", cout, cout, 3, 1, 1)
    for i in (0, 2, 4):
        add(f"mid_dilation.{i}", 128, 128, 1, 3, 3)
    for mod in ("backward_", "forward_"):
        p = f"feat_prop_module.deform_align.{mod}"
        add(p, 128, 256, 3, 3)
        add(p + ".conv_offset.0", 128, 384, 3, 3)
        add(p + ".conv_offset.2", 128, 128, 3, 3)
        add(p + ".conv_offset.4", 128, 128, 3, 3)
        add(p + ".conv_offset.6", 432, 128, 3, 3)
    add("feat_prop_module.backbone.backward_.0", 128, 256, 3, 3)
    add("feat_prop_module.backbone.backward_.2", 128, 128, 3, 3)
    add("feat_prop_module.backbone.forward_.0", 128, 384, 3, 3)
    add("feat_prop_module.backbone.forward_.2", 128, 128, 3, 3)
    add("feat_prop_module.fusion", 128, 256, 1, 1)
    add("decoder2.0", 128, 128, 3, 3)
    add("decoder2.2.conv", 64, 128, 3, 3)
    add("decoder1.0", 64, 64, 3, 3)
    add("decoder1.2.conv", 32, 64, 3, 3)
    add("upsample.0", 32, 32, 3, 3)
    add("upsample.2.conv", 2, 32, 3, 3)
    add("edgeDetector.projection.0", 16, 2, 3, 3)      # training-only head (:298-300); present in the checkpoint
    add("edgeDetector.mid_layer_1.0", 16, 16, 3, 3)
    add("edgeDetector.mid_layer_2.0", 16, 16, 3, 3)
    add("edgeDetector.out_layer", 1, 16, 1, 1)
    return spec


def make_rfc_state_dict(seed=0, profile="benign"):
    """Stand-in for weights/recurrent_flow_completion.pth (missing blob).  The offset head is NOT zero-initialised as
    in the reference's constructor (:27-28): offsets of a few pixels and non-trivial masks exercise the deformable
    sampling.  profile "heavy": Student-t weights, first conv x HEAVY_SCALE, the two output convs / HEAVY_SCALE."""
    _check_profile(profile)
    rng = np.random.default_rng(seed + 991)
    sd = {}
    for key, shape in rfc_state_dict_spec():
        if key.endswith("weight"):
            fan_in = int(np.prod(shape[1:]))
            gain = 1.2
            if "conv_offset.6" in key:
                gain = 0.4
            elif "conv2.0" in key or "backbone" in key or "fusion" in key:
                gain = 0.9
            if profile == "heavy" and key.startswith("downsample.0"):
                gain *= HEAVY_SCALE["rfc"]
            if profile == "heavy" and key.startswith("upsample.2.conv"):
                gain /= HEAVY_SCALE["rfc"]
            sd[key] = _draw(rng, shape, profile) * np.float32(gain / np.sqrt(fan_in))
        else:
            sd[key] = rng.standard_normal(shape).astype(np.float32) * np.float32(0.05)
    return sd


# ------------------------------------------------------------------------------------------------
# ProPainter InpaintGenerator (backend/inpaint/video/model/propainter.py:250-314)
# ------------------------------------------------------------------------------------------------
def propainter_valid_ind_rolled(window=(5, 9)):
    """SparseWindowAttention's buffer (sparse_transformer.py:141-154): indices of the rolled-window tokens that lie
    outside the current window, over the concatenation (top-left, top-right, bottom-left, bottom-right)."""
    wh, ww = window
    eh, ew = (wh + 1) // 2, (ww + 1) // 2
    m = np.ones((4, wh, ww), dtype=np.int64)
    m[0, :-eh, :-ew] = 0
    m[1, :-eh, ew:] = 0
    m[2, eh:, :-ew] = 0
    m[3, eh:, ew:] = 0
    return np.nonzero(m.reshape(-1))[0].astype(np.int64)


def propainter_state_dict_spec():
    spec = []

    def add(name, *shape):
        spec.append((name + ".weight", tuple(shape)))
        spec.append((name + ".bias", (shape[0],)))

    for i, (co, ci) in zip(range(0, 18, 2), ((64, 5), (64, 64), (128, 64), (256, 128), (384, 256), (512, 320), (384, 192), (256, 80), (128, 512))):
        add(f"encoder.layers.{i}", co, ci, 3, 3)
    add("decoder.0.conv", 128, 128, 3, 3)
    add("decoder.2", 64, 128, 3, 3)
    add("decoder.4.conv", 64, 64, 3, 3)
    add("decoder.6", 3, 64, 3, 3)
    add("ss.embedding", 512, 6272)
    add("sc.embedding", 6272, 512)
    add("sc.bias_conv", 128, 128, 3, 3)
    for mod in ("backward_1", "forward_1"):
        p = f"feat_prop_module.deform_align.{mod}"
        add(p, 128, 128, 3, 3)
        add(p + ".conv_offset.0", 128, 261, 3, 3)
        add(p + ".conv_offset.2", 128, 128, 3, 3)
        add(p + ".conv_offset.4", 128, 128, 3, 3)
        add(p + ".conv_offset.6", 432, 128, 3, 3)
    for mod in ("backward_1", "forward_1"):
        add(f"feat_prop_module.backbone.{mod}.0", 128, 258, 3, 3)
        add(f"feat_prop_module.backbone.{mod}.2", 128, 128, 3, 3)
    add("feat_prop_module.fuse.0", 128, 258, 3, 3)
    add("feat_prop_module.fuse.2", 128, 128, 3, 3)
    for i in range(8):
        p = f"transformers.transformer.{i}."
        spec.append((p + "attention.valid_ind_rolled", (148,)))
        for name in ("key", "query", "value", "proj"):
            add(p + "attention." + name, 512, 512)
        add(p + "attention.pool_layer", 512, 1, 4, 4)
        for name in ("norm1", "norm2"):
            add(p + name, 512)
        add(p + "mlp.fc1.0", 1960, 512)
        add(p + "mlp.fc2.1", 512, 1960)
    return spec


def make_propainter_state_dict(seed=0, profile="benign"):
    """Stand-in for weights/ProPainter.pth (missing blob): variance-preserving weights, non-trivial LayerNorm affine and
    deformable offsets (the reference zero-initialises the offset head, propainter.py:56-57).  profile "peaked": query / key gain
    x4; "heavy": Student-t weights, encoder.layers.0 x 30 and decoder.6 / 30, every block's fc1 x 1000 and fc2 / 1000 (HEAVY_SCALE)."""
    _check_profile(profile)
    rng = np.random.default_rng(seed + 313)
    sd = {}
    for key, shape in propainter_state_dict_spec():
        leaf = key.rsplit(".", 1)[1]
        if leaf == "valid_ind_rolled":
            sd[key] = propainter_valid_ind_rolled()
        elif ".norm" in key:
            sd[key] = (rng.uniform(0.7, 1.3, shape) if leaf == "weight" else rng.normal(0, 0.1, shape)).astype(np.float32)
        elif "pool_layer" in key:
            sd[key] = (np.full(shape, 1.0 / 16) if leaf == "weight" else np.zeros(shape)).astype(np.float32)
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:]))
            gain = 1.2
            if "conv_offset.6" in key:
                gain = 0.4
            elif "attention.query" in key or "attention.key" in key:
                gain = 1.6 * (4.0 if profile == "peaked" else 1.0)
            elif "mlp.fc2" in key or "attention.proj" in key or "sc.embedding" in key:
                gain = 0.7
            elif key.startswith("decoder.6"):
                gain = 0.7
            if profile == "heavy" and key.startswith("encoder.layers.0."):
                gain *= HEAVY_SCALE["propainter"]
            if profile == "heavy" and key.startswith("decoder.6"):
                gain /= HEAVY_SCALE["propainter"]
            if profile == "heavy" and "mlp.fc1" in key:
                gain *= HEAVY_SCALE["propainter_ffn"]
            if profile == "heavy" and "mlp.fc2" in key:
                gain /= HEAVY_SCALE["propainter_ffn"]
            sd[key] = _draw(rng, shape, profile) * np.float32(gain / np.sqrt(fan_in))
        else:
            sd[key] = rng.standard_normal(shape).astype(np.float32) * np.float32(0.05)
    return sd


# ------------------------------------------------------------------------------------------------
# big-LaMa generator (advimman/lama FFCResNetGenerator, ffc_resnet_075: ngf 64, 3 downsamplings, 18 FFC residual blocks at
# 512 channels with 75 % global channels, no LFU, sigmoid output).  The reference only ships it as the TorchScript blob
# big-lama.pt (a missing blob); key names follow the published module tree, `generator.` prefix optional.
# ------------------------------------------------------------------------------------------------
LAMA_NGF, LAMA_CL, LAMA_CG = 64, 128, 384          # bottleneck: 512 = 128 local + 384 global channels


def lama_state_dict_spec(n_blocks=18):
    spec = []

    def bn(name, c):
        spec.extend([(name + ".weight", (c,)), (name + ".bias", (c,)), (name + ".running_mean", (c,)), (name + ".running_var", (c,))])

    def conv(name, co, ci, k, bias=False):
        spec.append((name + ".weight", (co, ci, k, k)))
        if bias:
            spec.append((name + ".bias", (co,)))

    conv("model.1.ffc.convl2l", 64, 4, 7)
    bn("model.1.bn_l", 64)
    conv("model.2.ffc.convl2l", 128, 64, 3)
    bn("model.2.bn_l", 128)
    conv("model.3.ffc.convl2l", 256, 128, 3)
    bn("model.3.bn_l", 256)
    conv("model.4.ffc.convl2l", LAMA_CL, 256, 3)
    conv("model.4.ffc.convl2g", LAMA_CG, 256, 3)
    bn("model.4.bn_l", LAMA_CL)
    bn("model.4.bn_g", LAMA_CG)
    for i in range(n_blocks):
        for cv in ("conv1", "conv2"):
            p = f"model.{5 + i}.{cv}"
            conv(p + ".ffc.convl2l", LAMA_CL, LAMA_CL, 3)
            conv(p + ".ffc.convl2g", LAMA_CG, LAMA_CL, 3)
            conv(p + ".ffc.convg2l", LAMA_CL, LAMA_CG, 3)
            conv(p + ".ffc.convg2g.conv1.0", LAMA_CG // 2, LAMA_CG, 1)
            bn(p + ".ffc.convg2g.conv1.1", LAMA_CG // 2)
            conv(p + ".ffc.convg2g.fu.conv_layer", LAMA_CG, LAMA_CG, 1)
            bn(p + ".ffc.convg2g.fu.bn", LAMA_CG)
            conv(p + ".ffc.convg2g.conv2", LAMA_CG, LAMA_CG // 2, 1)
            bn(p + ".bn_l", LAMA_CL)
            bn(p + ".bn_g", LAMA_CG)
    base = 5 + n_blocks + 1                          # ConcatTupleLayer sits at 5 + n_blocks
    for j, (ci, co) in enumerate(((512, 256), (256, 128), (128, 64))):
        spec.append((f"model.{base + 3 * j}.weight", (ci, co, 3, 3)))          # ConvTranspose2d: [in, out, kh, kw]
        spec.append((f"model.{base + 3 * j}.bias", (co,)))
        bn(f"model.{base + 3 * j + 1}", co)
    conv(f"model.{base + 10}", 3, 64, 7, bias=True)
    return spec


def make_lama_state_dict(seed=0, n_blocks=18):
    """Stand-in for big-lama.pt's generator weights: variance-preserving convs, non-trivial eval-mode BatchNorm statistics; the
    second FFC of every residual block is damped so that 18 blocks keep the activations O(1)."""
    rng = np.random.default_rng(seed + 777)
    sd = {}
    for key, shape in lama_state_dict_spec(n_blocks):
        leaf = key.rsplit(".", 1)[1]
        is_bn = len(shape) == 1 and leaf != "bias" or (leaf == "bias" and (key[:-5] + ".running_mean") in dict(lama_state_dict_spec(n_blocks)))
        if leaf == "running_mean":
            sd[key] = rng.normal(0, 0.1, shape).astype(np.float32)
        elif leaf == "running_var":
            sd[key] = rng.uniform(0.6, 1.4, shape).astype(np.float32)
        elif is_bn and leaf == "weight":
            damp = 0.22 if ".conv2.bn_" in key else 1.0
            sd[key] = (rng.uniform(0.8, 1.2, shape) * damp).astype(np.float32)
        elif is_bn and leaf == "bias":
            sd[key] = rng.normal(0, 0.05, shape).astype(np.float32)
        elif leaf == "weight":
            tr = len(shape) == 4 and key.count(".") == 2 and shape[0] > shape[1] and shape[2] == 3       # ConvTranspose2d
            fan_in = int(shape[0] * shape[2] * shape[3] / 4) if tr else int(np.prod(shape[1:]))
            gain = 1.0 if "fu.conv_layer" in key or "convg2g.conv2" in key else 1.3
            if tr:
                gain = 0.9
            if len(shape) == 4 and shape[0] == 3:
                gain = 2.2 if n_blocks >= 10 else 14.0       # logits of the sigmoid stay O(1)
            if "convl2l" in key or "convg2l" in key or "convl2g" in key:
                gain = 0.95                                  # two branches are summed before the BatchNorm
            sd[key] = rng.standard_normal(shape).astype(np.float32) * np.float32(gain / np.sqrt(max(fan_in, 1)))
        else:
            sd[key] = rng.standard_normal(shape).astype(np.float32) * np.float32(0.05)
    return sd


def make_det_weights(graph, seed=0):
    """Stand-in for the text detector's weights (inference.pdiparams is a missing blob): {parameter name: fp32 array} for every
    parameter of a loaded PIR program (tools/paddle_graph.load_graph), fan-in scaled convolutions, positive BatchNorm variances.
    Good for timing and plumbing; the parity tests use oracle/ppocr_det.synthetic_weights, which also calibrates the activations."""
    rng = np.random.default_rng(seed + 2718)
    role = {}
    for kind, ins, _, _ in graph.ops:
        if kind == "batch_norm_":
            role[ins[1]], role[ins[2]], role[ins[3]], role[ins[4]] = "mean", "var", "scale", "shift"
        elif kind in ("conv2d", "depthwise_conv2d", "conv2d_transpose"):
            role[ins[1]] = "weight"
    out = {}
    for vid, (name, shape) in graph.params.items():
        r = role.get(vid, "bias" if len(shape) == 1 else "weight")
        if r == "weight" and len(shape) == 4:
            v = rng.standard_normal(shape) * (1.0 / np.sqrt(int(np.prod(shape[1:]))))
        elif r == "var":
            v = rng.uniform(0.5, 1.5, shape)
        elif r == "scale":
            v = rng.uniform(0.7, 1.3, shape)
        else:
            v = rng.normal(0, 0.1, shape)
        out[name] = np.asarray(v, dtype=np.float32)
    return out
