"""Frame sources / sinks for the plugins.

Codec work (cv2.VideoCapture, the ffmpeg libx264 pipe of backend/tools/video_io.py:50-103) is out
of scope for the accelerated path; the plugins only need ``read() -> (ok,
frame)`` and ``write(frame)``.  In-memory sources/sinks let the harness and tests feed synthetic
frames without a codec; a cv2-backed source is used when OpenCV is installed.
"""
import numpy as np


class ArrayVideo:
    """In-memory clip: uint8 [N,H,W,3] BGR.  Stands in for a video path."""

    def __init__(self, frames, fps=30.0):
        self.frames = frames
        self.fps = float(fps)
        self._pos = 0

    def info(self):
        n, h, w, _ = self.frames.shape
        return {"W_ori": int(w), "H_ori": int(h), "fps": self.fps, "len": int(n)}

    def read(self):
        if self._pos >= len(self.frames):
            return False, None
        f = self.frames[self._pos]
        self._pos += 1
        return True, f

    def release(self):
        pass


class ArrayWriter:
    """Collects written frames (the reference's writer converts non-uint8 by clip+cast, video_io.py:85-92)."""

    def __init__(self):
        self.frames = []

    def write(self, frame):
        if frame.dtype != np.uint8:
            frame = np.clip(frame, 0, 255).astype(np.uint8)
        self.frames.append(frame.copy())        # the caller may hand out views of a recycled (pinned) buffer

    def release(self):
        pass


class CountingWriter:
    """Sink that only counts (benchmarks: the encoder is out of scope)."""

    def __init__(self):
        self.count = 0
        self.checksum = 0

    def write(self, frame):
        self.count += 1
        self.checksum = (self.checksum + int(frame[::97, ::89].sum())) & 0xFFFFFFFF

    def release(self):
        pass


class Cv2Video:
    """cv2.VideoCapture-backed source (sttn_auto_inpaint.py:168-180); needs opencv-python."""

    def __init__(self, path):
        try:
            import cv2
        except ImportError as e:
            raise RuntimeError("reading a video file needs opencv-python (cv2); pass an ArrayVideo to feed "
                               "decoded frames directly") from e
        self._cv2 = cv2
        self.cap = cv2.VideoCapture(path)

    def info(self):
        cv2 = self._cv2
        return {"W_ori": int(self.cap.get(cv2.CAP_PROP_FRAME_WIDTH) + 0.5),
                "H_ori": int(self.cap.get(cv2.CAP_PROP_FRAME_HEIGHT) + 0.5),
                "fps": self.cap.get(cv2.CAP_PROP_FPS),
                "len": int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT) + 0.5)}

    def read(self):
        return self.cap.read()

    def release(self):
        self.cap.release()


def open_video(video):
    """A fresh reader positioned at frame 0 (the reference re-opens the file for every pass over the video)."""
    if isinstance(video, ArrayVideo):
        return ArrayVideo(video.frames, video.fps)
    return video if hasattr(video, "read") and hasattr(video, "info") else Cv2Video(video)
