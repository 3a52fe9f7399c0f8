"""Enum names and values of backend/tools/constant.py:4-19 (the --inpaint-mode plugin selector)."""
from enum import Enum, unique


@unique
class InpaintMode(Enum):
    STTN_AUTO = "sttn-auto"
    STTN_DET = "sttn-det"
    LAMA = "lama"
    PROPAINTER = "propainter"
    OPENCV = "opencv"


@unique
class SubtitleDetectMode(Enum):
    PP_OCRv5_MOBILE = "PP_OCRv5_MOBILE"
    PP_OCRv5_SERVER = "PP_OCRv5_SERVER"
