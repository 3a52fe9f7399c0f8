"""--inpaint-mode opencv (reference backend/inpaint/opencv_inpaint.py:1-16): cv2.inpaint on the CPU, one frame at a time.

Not on the accelerated path (out of scope as a kernel): the mode is a pass-through to OpenCV's own
Telea inpainting, exactly the call the reference makes, and exists only where opencv-python does.  Without cv2 the command line
refuses the mode (tools/args_handler.py) instead of failing after the detector pass."""


def available():
    try:
        import cv2  # noqa: F401
        return True
    except ImportError:
        return False


class OpenCVInpaint:
    def __init__(self):
        import cv2

        self._cv2 = cv2

    def inpaint(self, frame, mask):
        return self._cv2.inpaint(frame, mask, 3, self._cv2.INTER_LINEAR)      # opencv_inpaint.py:9 (flag value 1 = INPAINT_TELEA)

    def __call__(self, frames, mask):
        return [self.inpaint(frame, mask) for frame in frames]
