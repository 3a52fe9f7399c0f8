"""Detector quads -> axis-aligned boxes (reference backend/tools/ocr.py:1-20)."""


def get_coordinates(dt_box):
    """[[p1,p2,p3,p4], ...] (clockwise from top-left) -> [(xmin, xmax, ymin, ymax)]: the INNER box of each quad
    (max of the two left x, min of the two right x, max of the two top y, min of the two bottom y), ints truncated."""
    boxes = []
    if isinstance(dt_box, list):
        for quad in dt_box:
            (x1, y1), (x2, y2), (x3, y3), (x4, y4) = [(int(p[0]), int(p[1])) for p in list(quad)[:4]]
            boxes.append((max(x1, x4), min(x2, x3), max(y1, y2), min(y3, y4)))
    return boxes
