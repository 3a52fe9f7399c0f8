"""Command line of backend/main.py (reference backend/tools/args_handler.py:6-30): same flags,
same defaults, --inpaint-mode parsed into the InpaintMode enum."""
import argparse

from .constant import InpaintMode


def build_parser():
    parser = argparse.ArgumentParser(description="Video Subtitle Remover Command Line Tool")
    parser.add_argument("--input", "-i", required=True, type=str, help="Input video file path")
    parser.add_argument("--output", "-o", required=False, type=str, default=None,
                        help="Output video file path (optional)")
    parser.add_argument("--subtitle-area-coords", "-c", action="append", nargs=4, type=int,
                        metavar=("YMIN", "YMAX", "XMIN", "XMAX"),
                        help="Subtitle area coordinates (ymin ymax xmin xmax). Can be specified multiple times "
                             "for multiple areas.")
    parser.add_argument("--inpaint-mode", type=str, default="sttn-auto",
                        choices=[mode.name.lower().replace("_", "-") for mode in InpaintMode],
                        help="Inpaint mode, default is sttn-auto")
    return parser


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    args.inpaint_mode = InpaintMode[args.inpaint_mode.replace("-", "_").upper()]
    if args.inpaint_mode == InpaintMode.OPENCV:
        from ..inpaint import opencv_inpaint

        if not opencv_inpaint.available():
            # refused here, before any pass over the video: the mode is cv2.inpaint on the CPU (reference opencv_inpaint.py:9), which
            # this build neither accelerates nor re-implements
            build_parser().error("--inpaint-mode opencv is OpenCV's own CPU inpainting (cv2.inpaint): it needs opencv-python, which is "
                                 "not installed; the MI355X modes are sttn-auto, sttn-det, lama and propainter")
    if args.subtitle_area_coords is None:
        args.subtitle_area_coords = []
    return args
