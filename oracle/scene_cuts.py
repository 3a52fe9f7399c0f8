"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the scene-cut pass of --inpaint-mode propainter.

Follows the reference's vendored PySceneDetect: backend/tools/subtitle_detect.py:158-170 (get_scene_div_frame_no),
backend/scenedetect/scene_manager.py:132-148 (compute_downscale_factor), :499-504 (cv2.resize of every frame, INTER_LINEAR),
backend/scenedetect/detectors/content_detector.py:28-35 (_mean_pixel_distance), :138-172 (_calculate_frame_score, default
weights 1,1,1,0), :174-208 (process_frame: threshold 27.0, min_scene_len 15), scene_manager.py:183-197,712-719 (cuts -> scenes).

cv2 is absent from image and mount: cvtColor(BGR2HSV) on uint8 is restated from OpenCV 4.11's published integer algorithm
(modules/imgproc/src/color_hsv.simd.hpp, RGB2HSV_b: hsv_shift 12, sdiv_table / hdiv_table180) -- PARITY UNPINNED for that
operator; tests/test_scene_cuts.py holds hand-derived known answers (primaries, greys, the documented H/2 S*255 V*255 mapping).
"""
import numpy as np

from . import cv2_restate


def _tables():
    i = np.arange(1, 256, dtype=np.float64)
    sdiv = np.zeros(256, np.int64)
    hdiv = np.zeros(256, np.int64)
    sdiv[1:] = np.rint((255 << 12) / (1.0 * i)).astype(np.int64)          # saturate_cast<int>(double) = cvRound: half to even
    hdiv[1:] = np.rint((180 << 12) / (6.0 * i)).astype(np.int64)
    return sdiv, hdiv


_SDIV, _HDIV = _tables()


def bgr2hsv_u8(img):
    """cv2.cvtColor(img, cv2.COLOR_BGR2HSV) for uint8 [..., 3]"""
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(b, np.maximum(g, r))
    vmin = np.minimum(b, np.minimum(g, r))
    d = v - vmin
    s = (d * _SDIV[v] + (1 << 11)) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * d, r - g + 4 * d))
    h = (h * _HDIV[d] + (1 << 11)) >> 12                                   # arithmetic shift, as the C++ does on negative values
    h = np.where(h < 0, h + 180, h)
    return np.stack([np.clip(h, 0, 255), s, v], axis=-1).astype(np.uint8)


def downscale_size(W, H):
    f = 1 if W < 256 else W // 256
    return (W, H, f) if f <= 1 else (round(W / f), round(H / f), f)


def frame_sums(frames):
    """[n,H,W,3] u8 BGR -> int64 [n-1,3]: per-plane sums of |HSV(frame i+1) - HSV(frame i)| after the down-scaling"""
    n, H, W, _ = frames.shape
    w, h, f = downscale_size(W, H)
    hsv = [bgr2hsv_u8(cv2_restate.resize_linear(fr, (w, h)) if f > 1 else fr).astype(np.int64) for fr in frames]
    return np.array([[np.abs(hsv[i + 1][..., c] - hsv[i][..., c]).sum() for c in range(3)] for i in range(n - 1)], dtype=np.int64).reshape(n - 1, 3)


def scores_from_sums(sums, npix):
    """content_val of frames 1..n-1 (frame 0 scores 0.0): mean of the three per-plane mean distances"""
    comp = sums.astype(np.float64) / float(npix)
    return [float(sum(c * w for c, w in zip(list(row) + [0.0], (1.0, 1.0, 1.0, 0.0))) / 3.0) for row in comp]


def cuts_from_scores(scores, threshold=27.0, min_scene_len=15):
    """0-based frame numbers where a new scene starts; scores[i] belongs to frame i + 1"""
    cuts, last = [], 0
    for k, sc in enumerate(scores):
        frame_num = k + 1
        if sc >= threshold and frame_num - last >= min_scene_len:
            cuts.append(frame_num)
            last = frame_num
    return cuts


def scene_div_frame_no(frames):
    """SubtitleDetect.get_scene_div_frame_no: start.frame_num + 1 of every scene that does not start at frame 0"""
    n, H, W, _ = frames.shape
    if n < 2:
        return []
    w, h, _ = downscale_size(W, H)
    return [c + 1 for c in cuts_from_scores(scores_from_sums(frame_sums(frames), w * h))]
