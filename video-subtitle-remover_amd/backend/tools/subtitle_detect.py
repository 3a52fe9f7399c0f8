"""Temporal bookkeeping of the detector modes (reference backend/tools/subtitle_detect.py), detector injected.

The PP-OCRv5 network itself (paddleocr.TextDetection, :41-54) is a separate row of the scope table (its weights are
missing from the reference mount); everything around it -- sampling step, containment filter, gap filling, region
unification, interval construction -- is plain integer logic that decides which frames get which mask, so it is kept
verbatim in behaviour and pinned by tests/golden/bookkeeping.json (generated from the reference's own functions).
`text_detector.predict(img)` must return what paddleocr returns: an iterable of {'dt_polys': ndarray [k,4,2]}.
"""
from ..config import config
from .inpaint_tools import is_frame_number_in_ab_sections
from .ocr import get_coordinates
from .video_io import open_video


class SubtitleDetect:
    SAMPLE_STEP = 3

    def __init__(self, video_path, sub_areas=None, text_detector=None):
        self.video_path = video_path
        self.sub_areas = sub_areas if sub_areas is not None else []
        self.text_detector = text_detector
        self._init_sample_step()

    def _init_sample_step(self):
        """:29-39 -- keep at least ~8 samples per second."""
        fps = open_video(self.video_path).info()["fps"] if self.video_path is not None else 30
        self.SAMPLE_STEP = 4 if fps >= 60 else (3 if fps >= 30 else 2)

    def detect_subtitle(self, img):
        """:56-82 -- boxes of one frame, kept only if fully inside a user sub-area (ymin,ymax,xmin,xmax)."""
        if self.text_detector is None:
            raise RuntimeError("no text detector configured (PP-OCRv5 weights are not part of the reference mount)")
        return self._keep_inside(self.text_detector.predict(img))

    def _keep_inside(self, results):
        kept = []
        areas = self.sub_areas
        for res in results:
            polys = res["dt_polys"]
            if polys is None or len(polys) == 0:
                continue
            coords = get_coordinates(polys.tolist() if hasattr(polys, "tolist") else list(polys))
            if not coords:
                continue
            if not areas:
                kept.extend(coords)
                continue
            for xmin, xmax, ymin, ymax in coords:
                for s_ymin, s_ymax, s_xmin, s_xmax in areas:
                    if s_xmin <= xmin and xmax <= s_xmax and s_ymin <= ymin and ymax <= s_ymax:
                        kept.append((xmin, xmax, ymin, ymax))
                        break
        return kept

    def find_subtitle_frame_no(self, sub_remover=None, clip=None):
        """:84-132 -- {frame_no (1-based): [boxes]}: detect every SAMPLE_STEP-th frame, fill gaps <= 2 steps, unify.
        clip: the decoded video resident in HBM (tools/resident.ResidentClip): the sampled frames are taken from it instead of a
        second decoding pass over the file."""
        if clip is not None:
            return self._find_resident(sub_remover, clip)
        reader = open_video(self.video_path)
        sampled = {}
        frame_no = 0
        ab = sub_remover.ab_sections if sub_remover is not None else None
        # the sampled frames are independent: a detector with predict_batch (the MI355X one) takes several per forward
        batch = getattr(self.text_detector, "batch_size", 1) if hasattr(self.text_detector, "predict_batch") else 1
        wait = []

        def flush():
            if not wait:
                return
            if len(wait) == 1 or batch <= 1:
                results = [self.text_detector.predict(f) for _, f in wait]
            else:
                results = [[r] for r in self.text_detector.predict_batch([f for _, f in wait])]
            for (no, _), res in zip(wait, results):
                boxes = self._keep_inside(res)
                if len(boxes) > 0:
                    sampled[no] = boxes
            wait.clear()

        if self.text_detector is None:
            raise RuntimeError("no text detector configured (PP-OCRv5 weights are not part of the reference mount)")
        while True:
            ok, frame = reader.read()
            if not ok:
                break
            frame_no += 1
            if not is_frame_number_in_ab_sections(frame_no - 1, ab):
                continue
            if (frame_no - 1) % self.SAMPLE_STEP == 0 or self.SAMPLE_STEP <= 1:
                wait.append((frame_no, frame if frame.flags.owndata else frame.copy()))
                if len(wait) >= max(1, batch):
                    flush()
        flush()
        reader.release()
        return self.fill_and_unify(sampled)

    def _find_resident(self, sub_remover, clip):
        if self.text_detector is None:
            raise RuntimeError("no text detector configured (PP-OCRv5 weights are not part of the reference mount)")
        import torch

        ab = sub_remover.ab_sections if sub_remover is not None else None
        nos = [no for no in range(1, len(clip) + 1)
               if is_frame_number_in_ab_sections(no - 1, ab) and ((no - 1) % self.SAMPLE_STEP == 0 or self.SAMPLE_STEP <= 1)]
        from . import batch_lanes

        batch = max(1, getattr(self.text_detector, "batch_size", 1))
        on_device = hasattr(self.text_detector, "predict_batch_device")
        parts = [nos[s:s + batch] for s in range(0, len(nos), batch)]

        def detect(detector, part):
            idx = torch.tensor([no - 1 for no in part], dtype=torch.int64, device=clip.frames.device)
            if on_device:
                return [[r] for r in detector.predict_batch_device(clip.frames[idx])]
            return [detector.predict(f) for f in clip.frames[idx].cpu().numpy()]       # an injected detector with the reference's host signature

        # the sampled frames are independent: VSR_DET_LANES detectors (own runner, own stream, own host thread) share the batches
        # (default 2 since round 4, VSR_DET_LANES=1 = this thread alone; the results do not depend on it: tests/test_batch_lanes.py,
        # tests/test_gpu_ocr_det.py::test_detector_lanes_on_the_device)
        if not hasattr(self, "_det_lanes"):
            self._det_lanes = {}
        detectors = batch_lanes.lane_plugins(self.text_detector,
                                             batch_lanes.lanes_from_env("VSR_DET_LANES", getattr(self, "det_lanes_default", None)) if on_device else 1,
                                             self._det_lanes)
        sampled = {}
        for part, results in zip(parts, batch_lanes.run_map(parts, detectors, detect, clip.frames.device)):
            for no, res in zip(part, results):
                boxes = self._keep_inside(res)
                if len(boxes) > 0:
                    sampled[no] = boxes
        return self.fill_and_unify(sampled)

    def fill_and_unify(self, sampled):
        """Phase 2 of find_subtitle_frame_no (:112-131), separated so that it can be tested without a video."""
        filled = {}
        nos = sorted(sampled)
        max_gap = self.SAMPLE_STEP * 2
        for f, nxt in zip(nos, nos[1:]):
            filled[f] = sampled[f]
            if nxt - f <= max_gap:
                for g in range(f + 1, nxt):
                    filled[g] = sampled[f]
        if nos:
            filled[nos[-1]] = sampled[nos[-1]]
        filled = self.unify_regions(filled)
        return {k: v for k, v in filled.items() if len(v) > 0}

    @staticmethod
    def are_similar(region1, region2):
        tx = config.subtitleAreaPixelToleranceXPixel.value
        ty = config.subtitleAreaPixelToleranceYPixel.value
        return (abs(region1[0] - region2[0]) <= tx and abs(region1[1] - region2[1]) <= tx
                and abs(region1[2] - region2[2]) <= ty and abs(region1[3] - region2[3]) <= ty)

    def unify_regions(self, raw_regions):
        """:181-215 -- a box similar to the same-index box of the previous key takes that box's coordinates."""
        if len(raw_regions) == 0:
            return raw_regions
        keys = sorted(raw_regions)
        unified = {keys[0]: raw_regions[keys[0]]}
        prev = keys[0]
        for key in keys[1:]:
            row = []
            for idx, region in enumerate(raw_regions[key]):
                std = unified[prev][idx] if idx < len(unified[prev]) else None
                row.append(std if std and self.are_similar(region, std) else region)
            unified[key] = row
            prev = key
        return {k: unified[k] for k in keys}

    @staticmethod
    def find_continuous_ranges(subtitle_frame_no_box_dict):
        nums = sorted(subtitle_frame_no_box_dict)
        ranges, start = [], nums[0]
        for a, b in zip(nums, nums[1:]):
            if b - a != 1:
                ranges.append((start, a))
                start = b
        ranges.append((start, nums[-1]))
        return ranges

    @staticmethod
    def find_continuous_ranges_with_same_mask(subtitle_frame_no_box_dict):
        """:238-258 -- a run also ends where the box list changes between consecutive frames."""
        nums = sorted(subtitle_frame_no_box_dict)
        ranges, start = [], nums[0]
        for a, b in zip(nums, nums[1:]):
            if b - a != 1 or subtitle_frame_no_box_dict[b] != subtitle_frame_no_box_dict[a]:
                ranges.append((start, a))
                start = b
        ranges.append((start, nums[-1]))
        return ranges

    @staticmethod
    def get_scene_div_frame_no(v_path, device=0, clip=None):
        """subtitle_detect.py:158-170 -- frame numbers (1-based) where a new scene starts; the ContentDetector pass runs on the GPU
        (tools/scene_detect.py), on the HBM-resident clip when there is one"""
        from . import scene_detect

        return scene_detect.get_scene_div_frame_no(v_path, device=device, clip=clip)

    @staticmethod
    def split_range_by_scene(intervals, points):
        """tools/subtitle_detect.py:135-155: cut every (start, end) interval at the scene-change frame numbers inside it."""
        points = sorted(points)
        out = []
        for start, end in intervals:
            for p in [q for q in points if start <= q <= end]:
                if start < p:
                    out.append((start, p - 1))
                start = p
            out.append((start, end))
        return out

    @staticmethod
    def filter_and_merge_intervals(intervals, target_length):
        """:261-293 -- single-frame intervals grow to target_length where neighbours allow; short touching ones merge."""
        if not intervals:
            return []
        intervals = sorted(intervals, key=lambda t: t[0])
        grown = []
        for i, (s, e) in enumerate(intervals):
            if s == e:
                prev_end = grown[-1][1] if grown else float("-inf")
                next_start = intervals[i + 1][0] if i + 1 < len(intervals) else float("inf")
                half = (target_length - 1) // 2
                ns, ne = max(s - half, prev_end + 1), min(s + half, next_start - 1)
                if ne < ns:
                    ns, ne = s, s
                grown.append((ns, ne))
            else:
                grown.append((s, e))
        merged = [grown[0]]
        for s, e in grown[1:]:
            ls, le = merged[-1]
            if (s <= le or s == le + 1) and ((e - s + 1) < target_length or (le - ls + 1) < target_length):
                merged[-1] = (ls, max(le, e))
            else:
                merged.append((s, e))
        return merged
