"""Scene-cut pass of --inpaint-mode propainter on the MI355X.

Mirror of what the reference gets from its vendored PySceneDetect: SubtitleDetect.get_scene_div_frame_no
(backend/tools/subtitle_detect.py:158-170) = scene_detect(path, ContentDetector()) -> first frame (1-based) of every scene
but the first.  Per frame the reference down-scales (scene_manager.py:132-148 factor = W // 256, :499-504 cv2.resize INTER_LINEAR),
converts BGR->HSV, and scores the mean absolute difference of the three planes against the previous frame
(content_detector.py:28-35, :138-172); a cut needs score >= 27.0 and >= 15 frames since the last one (:174-208).
The pixel work (resize, HSV, |difference| sums: integers, bit-exact) runs in HIP kernels over batches of frames resident in
HBM; only the three sums per frame pair come back.  No CPU path.
"""
import ctypes as C

import numpy as np
import torch

from ..._lib import check, lib
from .video_io import open_video

THRESHOLD = 27.0          # ContentDetector defaults (content_detector.py:104-105)
MIN_SCENE_LEN = 15
DEFAULT_MIN_WIDTH = 256   # scene_manager.py:108


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def compute_downscale_factor(frame_width, effective_width=DEFAULT_MIN_WIDTH):      # scene_manager.py:132-148
    return 1 if frame_width < effective_width else frame_width // effective_width


class ContentDetector:
    """threshold / min_scene_len as the reference's class; frame_sums() is the device part, process() the cut logic."""

    def __init__(self, threshold=THRESHOLD, min_scene_len=MIN_SCENE_LEN, device=0, batch_frames=64):
        if not torch.cuda.is_available():
            raise RuntimeError("scene detection runs on the MI355X path only: no HIP device available")
        self.threshold, self.min_scene_len = float(threshold), int(min_scene_len)
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.batch_frames = int(batch_frames)
        self._tables = {}

    def _resize_tables(self, H, W, h, w):
        key = (H, W, h, w)
        if key not in self._tables:
            tabs = []
            for ssize, dsize, clamp in ((W, w, 1), (H, h, 0)):
                ofs, ic, fc = np.zeros(dsize, np.int32), np.zeros(2 * dsize, np.int16), np.zeros(2 * dsize, np.float32)
                check(lib.vsr_cv2_linear_tables(ssize, dsize, clamp, ofs.ctypes.data_as(C.c_void_p), ic.ctypes.data_as(C.c_void_p),
                                                fc.ctypes.data_as(C.c_void_p)))
                tabs += [torch.from_numpy(ofs).to(self.device), torch.from_numpy(ic).to(self.device)]
            self._tables[key] = tabs
        return self._tables[key]

    def _buffers(self, H, W):
        f = compute_downscale_factor(W)
        w, h = (round(W / f), round(H / f)) if f > 1 else (W, H)
        key = ("buf", H, W)
        if key not in self._tables:
            B = self.batch_frames
            self._tables[key] = (torch.empty((B + 1, h, w, 3), dtype=torch.uint8, device=self.device),   # slot 0 = last frame of the previous batch
                                 torch.empty((B, h, w, 3), dtype=torch.uint8, device=self.device) if f > 1 else None,
                                 torch.empty((B, 3), dtype=torch.int64, device=self.device))
        return (f, w, h) + self._tables[key]

    def device_batch(self, src, have_prev):
        """src: uint8 [n,H,W,3] BGR on the device, n <= batch_frames -> device int64 [n-1 (+1 with a carried frame), 3] sums (a view
        of a reused buffer); afterwards the batch's last frame is the carried one"""
        n, H, W, _ = src.shape
        f, w, h, hsv, small, sums_dev = self._buffers(H, W)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if f > 1:
            xofs, ialpha, yofs, ibeta = self._resize_tables(H, W, h, w)
            check(lib.vsr_launch_resize_u8(_p(src), H * W * 3, W * 3, W, H, _p(small), w, h, n, 3, None, _p(xofs), _p(ialpha), _p(yofs),
                                           _p(ibeta), stream))
            src = small
        check(lib.vsr_launch_bgr2hsv_u8(_p(src), _p(hsv[1:]), n * h * w, stream))
        first = 0 if have_prev else 1
        npairs = n - 1 + (1 if have_prev else 0)
        if npairs > 0:
            check(lib.vsr_launch_absdiff_sums_u8x3(_p(hsv[first:]), npairs, h * w, _p(sums_dev), stream))
        hsv[0].copy_(hsv[n])
        return sums_dev[:npairs]

    def frame_sums(self, frames_iter, H, W):
        """frames (HxWx3 uint8 BGR, host) in video order -> (int64 [n-1, 3] sums of |HSV difference| per plane, pixels per frame)"""
        _, w, h = self._buffers(H, W)[:3]
        out, have_prev, batch = [], False, []

        def flush():
            nonlocal have_prev
            if not batch:
                return
            with torch.cuda.device(self.device):
                src = torch.from_numpy(np.ascontiguousarray(np.stack(batch))).to(self.device)
                sums = self.device_batch(src, have_prev)
                if sums.shape[0]:
                    out.append(sums.cpu().numpy().copy())
                torch.cuda.synchronize(self.device)
            have_prev = True
            batch.clear()

        for fr in frames_iter:
            if fr.shape != (H, W, 3) or fr.dtype != np.uint8:
                raise ValueError(f"frame of shape {fr.shape} / {fr.dtype}, expected ({H}, {W}, 3) uint8")
            batch.append(fr)
            if len(batch) == self.batch_frames:
                flush()
        flush()
        sums = np.concatenate(out) if out else np.zeros((0, 3), np.int64)
        return sums, h * w

    def process(self, sums, npix):
        """-> 0-based numbers of the frames that start a new scene.  Same float arithmetic as _mean_pixel_distance /
        _calculate_frame_score: each plane's sum / num_pixels, weighted (1, 1, 1, 0) sum / 3."""
        cuts, last = [], 0                       # _last_scene_cut starts at the first frame number (content_detector.py:196-197)
        for k in range(sums.shape[0]):
            comps = [float(int(s) / float(npix)) for s in sums[k]] + [0.0]
            score = sum(c * wgt for c, wgt in zip(comps, (1.0, 1.0, 1.0, 0.0))) / 3.0
            frame_num = k + 1
            if score >= self.threshold and frame_num - last >= self.min_scene_len:
                cuts.append(frame_num)
                last = frame_num
        return cuts


def get_scene_div_frame_no(video, device=0, detector=None, clip=None):
    """SubtitleDetect.get_scene_div_frame_no (subtitle_detect.py:158-170): `start.frame_num + 1` of every detected scene that
    does not start at frame 0.  `video`: a path or frame source accepted by video_io.open_video; clip: the same video resident in
    HBM (tools/resident.ResidentClip) -- the kernels then read it where it is, no third decoding pass."""
    det = detector if detector is not None else ContentDetector(device=device)
    if clip is not None:
        n, H, W, _ = clip.frames.shape
        npix = None
        out, have_prev = [], False
        with torch.cuda.device(det.device):
            _, w, h = det._buffers(H, W)[:3]
            for s in range(0, n, det.batch_frames):
                sums = det.device_batch(clip.frames[s:s + det.batch_frames], have_prev)
                if sums.shape[0]:
                    out.append(sums.cpu().numpy().copy())
                have_prev = True
        sums = np.concatenate(out) if out else np.zeros((0, 3), np.int64)
        return [c + 1 for c in det.process(sums, h * w)]
    reader = open_video(video)
    info = reader.info()

    def frames():
        while True:
            ok, fr = reader.read()
            if not ok:
                return
            yield fr

    try:
        sums, npix = det.frame_sums(frames(), info["H_ori"], info["W_ori"])
    finally:
        reader.release()
    return [c + 1 for c in det.process(sums, npix)]
