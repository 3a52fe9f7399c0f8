"""The selector enums the command line and the config speak (reference backend/tools/constant.py:4-19): member names and
values are the reference's, because `--inpaint-mode` parses `InpaintMode[NAME]` and the config file stores the values."""
from enum import Enum

# --inpaint-mode plugin selector: command-line spelling -> member name
_INPAINT_MODES = ("sttn-auto", "sttn-det", "lama", "propainter", "opencv")
InpaintMode = Enum("InpaintMode", {v.upper().replace("-", "_"): v for v in _INPAINT_MODES}, module=__name__)

# text detection program (PP-OCRv5 mobile / server); the value is the member name
SubtitleDetectMode = Enum("SubtitleDetectMode", {n: n for n in ("PP_OCRv5_MOBILE", "PP_OCRv5_SERVER")}, module=__name__)
