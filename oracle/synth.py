"""Seeded synthetic clips for parity tests and bench.py.

A smooth low-frequency background that translates a few pixels per frame (so temporal
attention has signal), plus white "subtitle" glyph blocks inside the box.  uint8 BGR frames.
"""
import numpy as np


def make_clip(n, H, W, box, seed=0):
    """box = (ymin, ymax, xmin, xmax) of the subtitle area (CLI order, args_handler.py:19)."""
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
        while x + gh_px < xmax - 8:
            wpx = int(grng.integers(gh_px // 2, gh_px + 1))
            if grng.random() < 0.8:
                img[gy:gy + gh_px, x:x + wpx] = 16
                img[gy + 2:gy + gh_px - 2, x + 2:x + wpx - 2] = 250
            x += wpx + max(gh_px // 4, 2)
        frames[i] = img
    return frames
