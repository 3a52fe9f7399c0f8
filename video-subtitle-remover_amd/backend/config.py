"""Dependency-free stand-in for the reference's backend/config.py.

The reference builds a qfluentwidgets.QConfig singleton persisted to config/config.json
(backend/config.py:21-113); the hot path only reads the handful of values below through
``config.<name>.value`` and ``config.getSttnMaxLoadNum()``, so the same attribute names and
defaults are kept and everything else (UI language, window geometry, update URLs) is out of scope.
"""
from .tools.constant import InpaintMode, SubtitleDetectMode

VERSION = "1.4.0"


class _Item:
    def __init__(self, value):
        self.value = value


class Config:
    def __init__(self):
        self.inpaintMode = _Item(InpaintMode.STTN_AUTO)                         # config.py:53
        self.subtitleDetectMode = _Item(SubtitleDetectMode.PP_OCRv5_SERVER)      # config.py:55
        self.subtitleYXAxisDifferencePixel = _Item(10)                          # config.py:59
        self.subtitleAreaDeviationPixel = _Item(10)                             # config.py:61
        self.subtitleAreaYAxisDifferencePixel = _Item(20)
        self.subtitleAreaPixelToleranceYPixel = _Item(20)                       # config.py:65
        self.subtitleAreaPixelToleranceXPixel = _Item(20)                       # config.py:66
        self.subtitleTimelineBackwardFrameCount = _Item(3)                      # config.py:67
        self.subtitleTimelineForwardFrameCount = _Item(3)                       # config.py:68
        self.sttnNeighborStride = _Item(5)
        self.sttnReferenceLength = _Item(10)
        self.sttnMaxLoadNum = _Item(50)
        self.propainterMaxLoadNum = _Item(70)                                   # config.py:100
        self.hardwareAcceleration = _Item(True)                                 # config.py:103

    def getSttnMaxLoadNum(self):
        """config.py:89-94: max(sttnMaxLoadNum, sttnNeighborStride * sttnReferenceLength)."""
        return max(self.sttnMaxLoadNum.value, self.sttnNeighborStride.value * self.sttnReferenceLength.value)


config = Config()
